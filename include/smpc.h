/*
 * smpc.h -- C ABI of the MI355X batched safe-MPC engine.
 *
 * This is the drop-in boundary for ONE path of idra-lab/safe-mpc: everything at and below
 * AbstractController.solve() (reference src/safe_mpc/controller.py:136-167), i.e. what the reference
 * reaches through acados_template.AcadosOcpSolver (controller.py:247) -- batched over independent OCP
 * instances.  Plain pointers and sizes only; no torch / numpy types.  The same structs are mirrored with
 * ctypes in safe_mpc_amd/problem.py and are also read by the test oracle (oracle/), which shares this
 * interface and nothing else with the product.
 *
 * Conventions
 *   - all arrays are row-major and laid out exactly like the reference's numpy arrays:
 *       x0 [B][nx], x_guess [B][N+1][nx], u_guess [B][N][nu], p [B][N+1][5]   (guess_acados.py:236,
 *       controller.py:147-156); nx = 2*nq, nu = nq, p = [ee_ref(3), alpha, flag] (controller.py:27-31).
 *   - every entry point returns 0 on success and a negative SMPC_E* code on API misuse / HIP failure.
 *     The numerical outcome of each instance is reported only through status[B], using the acados codes the
 *     reference tests against (0 ok, 1 NaN, 2 max-iter, 3 min-step, 4 QP failure; controller.py:125,158).
 *   - a handle is bound to one device and one stream, owns all device scratch, and is not thread-safe.
 */
#ifndef SMPC_H_
#define SMPC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMPC_ABI_VERSION 5

#define SMPC_MAX_NQ 7
#define SMPC_MAX_NX 14
#define SMPC_MAX_POINTS 12
#define SMPC_MAX_ROWS 12
#define SMPC_MAX_LAYERS 6
#define SMPC_MAX_N 63
#define SMPC_NP 5 /* per-node parameter vector length */

/* bounds with |value| >= SMPC_INF are treated as absent (the reference writes 1e6 for "no upper bound",
 * env_model.py:265-266, safe_set.py:104) */
#define SMPC_INF 1.0e5

/* error codes */
#define SMPC_OK 0
#define SMPC_EINVAL (-1)
#define SMPC_ENOMEM (-2)
#define SMPC_EHIP (-3)
#define SMPC_ESTATE (-4)

/* acados status codes reproduced per instance */
#define SMPC_STATUS_SUCCESS 0
#define SMPC_STATUS_NAN 1
#define SMPC_STATUS_MAXITER 2
#define SMPC_STATUS_MINSTEP 3
#define SMPC_STATUS_QP_FAILURE 4

/* One actuated revolute joint and the (lumped) link it moves.  Replaces the adam KinDynComputations model the
 * reference builds from the URDF (env_model.py:40-45).  Links reached through fixed / locked joints are lumped
 * into `mass, com, inertia` by the host (safe_mpc_amd/urdf.py). */
typedef struct {
    double R0[9];      /* parent-link frame -> joint frame rotation at q = 0, row-major (URDF origin rpy) */
    double p0[3];      /* ... and translation (URDF origin xyz) */
    double axis[3];    /* unit axis in the joint (= child link) frame */
    double mass;       /* lumped mass of the child link */
    double com[3];     /* lumped centre of mass, child-link frame */
    double inertia[6]; /* ixx ixy ixz iyy iyz izz about the COM, child-link axes */
    double q_min, q_max, v_max, tau_max; /* URDF <limit> (env_model.py:107-114) */
} smpc_joint;

/* A point rigidly attached to the child link of actuated joint `link` (link = -1: fixed in the world). */
typedef struct {
    int32_t link;
    int32_t reserved;
    double local[3];
} smpc_point;

/* kinds of collision rows (env_model.py:263-316) */
#define SMPC_ROW_SEG_FIXEDSEG 0 /* capsule-capsule, second capsule fixed (utils.py:94-113) */
#define SMPC_ROW_SEG_SEG 1      /* capsule-capsule, both on the robot */
#define SMPC_ROW_SEG_POINT 2    /* capsule-sphere (utils.py:115-118) */
#define SMPC_ROW_POINT_POINT 3  /* sphere-sphere on the EE point (env_model.py:300-301) */
#define SMPC_ROW_COORD 4        /* plane: coordinate of a robot point (env_model.py:286-287, utils.py:123-124) */

typedef struct {
    int32_t kind;
    int32_t pa, pb; /* robot points: segment A-B, or the single point pa */
    int32_t pc, pd; /* second robot segment (SEG_SEG) */
    int32_t axis;   /* COORD: 0/1/2 */
    double C[3];    /* fixed segment start / fixed point */
    double D[3];    /* fixed segment end */
    double len2;    /* SEG_POINT: capsule_length**2 used as denominator (utils.py:116) */
    double offset;  /* COORD: value = P[axis] - offset */
    double lb, ub;  /* lh, uh of this row */
} smpc_row;

#define SMPC_COST_ZERO 0  /* ZeroCost (cost_definition.py:34-46) */
#define SMPC_COST_REACH 1 /* ReachTargetEXT / ReachTargetNLS (cost_definition.py:61-100) */

#define SMPC_HESS_GAUSS_NEWTON 0 /* NONLINEAR_LS */
#define SMPC_HESS_EXACT 1        /* EXTERNAL + hessian_approx EXACT (cost_definition.py:18,100) */

#define SMPC_NN_NONE 0
#define SMPC_NN_TERMINAL 1 /* ST / HTWA / RealReceding: node N only (controller.py:332-357) */
#define SMPC_NN_ALL 2      /* Receding / constraint_everywhere: nodes 1..N, switched by p[4] (controller.py:411-442) */

typedef struct {
    int32_t abi_version;
    int32_t nq;
    int32_t N;
    int32_t n_points;
    int32_t n_rows;
    int32_t ee_point;          /* index of the EE point (env_model.py:92-95) */
    int32_t cost_kind;
    int32_t hessian;
    int32_t nn_mode;
    int32_t nn_dof;            /* n_dof_safe_set (config.yaml:11) */
    int32_t qp_max_iter;       /* qp_solver_iter_max (config.yaml:18) */
    int32_t rows_at_node0;     /* != 0: the collision rows are kept at node 0, as the reference does when --noise == 0
                                * (controller.py:77-79; removed for noisy runs, :69-73).  x_0 is pinned, so they are
                                * constants of the QP: a violated one makes the QP infeasible and the instance reports
                                * SMPC_STATUS_QP_FAILURE (the iterate is still returned, as acados does) */
    int32_t qp_stall_iters;    /* > 0: the IPM gives up (SMPC_STATUS_QP_FAILURE, the iterate is still returned) after this many
                                * CONSECUTIVE stalled iterations -- step length below 1/2 AND the complementarity not halved
                                * either (an iterate that meets the exit test is never a stall) -- or after 7/6 of that many in
                                * total (round 5: 24 -> 28; infeasible QPs whose complementarity falls in bursts reset the run
                                * and took up to 52 iterations, now 31, with no feasible QP more given up).  0 = off: an infeasible QP then runs
                                * until its step length underflows (40-90 iterations: what RealReceding's +-1e-3 tubes,
                                * controller.py:531-532, produce in about 1 % of its solves).  A stall is not a proof of
                                * infeasibility -- a feasible, degenerate QP can crawl for 20 iterations before it converges
                                * (tests/golden/c4_degenerate_start.npz) -- hence an option with a generous default where it is
                                * switched on (24 for 'real_receding' and for the backup OCP, problem.py) and none elsewhere */
    int32_t reserved_i0;
    double dt;                 /* config.yaml:7 */
    double Q, R;               /* config.yaml:35,39 */
    double cost_scale_stage;   /* factor on the cost Q|ee-ref|^2 + R|u|^2 whose derivatives smpc_node_eval reports: acados
                                * multiplies stage costs by dt and the terminal one by 1 [EXT-UNVERIFIED]; a NONLINEAR_LS cost
                                * is 1/2 |y|^2_W (cost_definition.py:61-81), i.e. a further factor 1/2 on both */
    double cost_scale_term;
    double lm_stage;           /* Levenberg-Marquardt added to every diagonal of the stage Hessian */
    double lm_term;
    double nn_eps;             /* config.yaml:48 */
    double nn_soft_e;          /* L1 slack weight on the terminal NN row (zl_e, controller.py:348-354); < 0 = hard */
    double nn_soft_run;        /* same for running nodes; < 0 = hard */
    double qp_tol;             /* IPM exit tolerance on the complementarity (mean lambda t) */
    double qp_tol_res;         /* ... and on the linear residuals (stationarity, dynamics, slack definitions); 0 = qp_tol.
                                * HPIPM's BALANCE mode (config.yaml:15) asks 1e-6 of the stationarity residual and 1e-8 of the
                                * others [EXT-UNVERIFIED]; the engine's default is 1e-8 for both */
    double qp_mu0;             /* IPM initial barrier */
    double gravity[3];
    double nn_mean[SMPC_MAX_NQ];
    double nn_std[SMPC_MAX_NQ];
    double x_lo[SMPC_MAX_NX];   /* lbx / ubx of nodes 1..N-1 (controller.py:49-51) */
    double x_hi[SMPC_MAX_NX];
    double x_lo_e[SMPC_MAX_NX]; /* lbx_e / ubx_e (controller.py:53-55, 300-306) */
    double x_hi_e[SMPC_MAX_NX];
    smpc_joint joints[SMPC_MAX_NQ];
    smpc_point points[SMPC_MAX_POINTS];
    smpc_row rows[SMPC_MAX_ROWS];
} smpc_problem_desc;

/* Per-(instance, node) linearisation record returned by smpc_eval_nodes; one per node k = 0..N.
 * Exposes what acados evaluates through the CasADi-generated functions (N2 in SURVEY section 2) so that each piece
 * can be compared with the oracle separately. */
typedef struct {
    double tau[SMPC_MAX_NQ];                       /* M(q)u + h(q,qd) (env_model.py:80-83) */
    double M[SMPC_MAX_NQ * SMPC_MAX_NQ];           /* dtau/du, row-major nq x nq (leading dim nq) */
    double dtau_dq[SMPC_MAX_NQ * SMPC_MAX_NQ];
    double dtau_dv[SMPC_MAX_NQ * SMPC_MAX_NQ];
    double ee[3];                                  /* t_glob (env_model.py:92-95) */
    double cost_grad_q[SMPC_MAX_NQ];               /* d/dq of Q*|ee - ref|^2 (unscaled) */
    double cost_hess_qq[SMPC_MAX_NQ * SMPC_MAX_NQ];/* exact or Gauss-Newton, unscaled, row-major */
    double row_val[SMPC_MAX_ROWS];                 /* collision rows */
    double row_grad[SMPC_MAX_ROWS * SMPC_MAX_NQ];  /* d row / dq, row-major n_rows x nq */
    double nn_val;                                 /* g(x,p) (safe_set.py:94), 0 if not evaluated */
    double nn_grad[SMPC_MAX_NX];                   /* dg/dx */
} smpc_node_eval;

typedef struct smpc_handle smpc_handle;

/* ---- lifetime ------------------------------------------------------------------------------------------------- */
/* replaces AcadosOcpSolver(ocp, json_file, generate, build) (controller.py:247); device = HIP device ordinal */
int smpc_create(const smpc_problem_desc* desc, int device, smpc_handle** out);
void smpc_destroy(smpc_handle* h);
int smpc_abi_version(void);
/* last error text of this handle (or of the failed smpc_create when h == NULL) */
const char* smpc_last_error(const smpc_handle* h);

/* replaces l4c.L4CasADi(model_net, device='cpu') + model_external_shared_lib_* (safe_set.py:89-94,
 * controller.py:344-346): fp32 weights W[l] is [dims[l+1]][dims[l]] row-major (torch nn.Linear.weight), b[l] is
 * [dims[l+1]]; activation between layers is GELU(tanh) (parser.py:99), none after the last.  Pointers may be host or
 * device memory (on_device != 0: e.g. torch.Tensor.data_ptr() of a ROCm tensor); the handle keeps its own copy. */
int smpc_set_mlp(smpc_handle* h, int nlayers, const int32_t* dims, const float* const* W, const float* const* b,
                 int on_device);
/* the activation between the layers: parser.py:95-102 (`act_fun` of config.yaml:67); GELU(tanh) unless set */
enum { SMPC_ACT_GELU_TANH = 0, SMPC_ACT_RELU = 1, SMPC_ACT_ELU = 2, SMPC_ACT_TANH = 3, SMPC_ACT_SILU = 4 };
int smpc_set_mlp_activation(smpc_handle* h, int act);

/* replaces ocp_solver.set_new_time_steps + update_qp_solver_cond_N (controller.py:208-209): change N without
 * re-creating; N <= SMPC_MAX_N */
int smpc_set_horizon(smpc_handle* h, int N);

/* Which form of the QP solve a handle launches -- the engine's counterpart of the reference's choice of QP back-end
 * (`ocp.solver_options.qp_solver`, controller.py:100-101: partial- or full-condensing HPIPM; `qp_solver_cond_N`, controller.py:209).
 * Same algorithm and the same result to rounding either way; only the mapping onto the GPU differs:
 *   SMPC_QP_AUTO (default)  by batch size: the latency form for small batches, the throughput form otherwise
 *   SMPC_QP_THROUGHPUT      k_qp_ipm: one wavefront per pair of instances
 *   SMPC_QP_LATENCY         k_qp_ipm_wg: one workgroup per instance, stage-parallel row work, recursions through LDS
 *                           (falls back to the throughput form when a horizon's factor blocks do not fit one CU's LDS)
 * (added in round 6; ABI version unchanged: no existing entry point or structure changed) */
enum { SMPC_QP_AUTO = -1, SMPC_QP_THROUGHPUT = 0, SMPC_QP_LATENCY = 1 };
int smpc_set_qp_mode(smpc_handle* h, int mode);

/* replaces ocp_solver.constraints_set(k,'lbx'/'ubx',v) for k >= 1 (controller.py:531-536).  lo/hi are [N+1][nx]
 * shared by all instances, or NULL to restore the descriptor's bounds. */
int smpc_set_stage_bounds(smpc_handle* h, const double* lo, const double* hi);

/* Per-instance variant for RealReceding's state tube (controller.py:530-536: node r of instance b is boxed to
 * x_guess[b][r+1] +- 1e-3 with its own r): lo/hi are [B][N+1][nx]; they apply to the next smpc_solve_batch calls with the
 * same B until cleared with lo = hi = NULL.  Pointers follow on_device like smpc_solve_batch; the handle keeps a copy. */
int smpc_set_instance_bounds(smpc_handle* h, int B, const double* lo, const double* hi, int on_device);

/* replaces ocp_solver.cost_set(k,'zl'/'zu',v) (controller.py:455-468, 526-527): L1 penalty of the slack on the safe-set row of
 * node k, for the nodes where the formulation made that row soft (nn_soft_e / nn_soft_run >= 0; a hard row has no slack and
 * acados' arrays for it are empty).  zl is [N+1] host doubles shared by all instances (entry 0 unused); NULL restores the
 * descriptor's weights. */
int smpc_set_slack_weights(smpc_handle* h, const double* zl);

/* ---- the hot path --------------------------------------------------------------------------------------------- */
/* One SQP-RTI solve of B independent OCPs: replaces reset / constraints_set(0,lbx|ubx,x0) / set(i,x|u|p) / solve /
 * get(i,x|u) of controller.py:141-164 for B instances in one call.
 *   x0 [B][nx], xg [B][N+1][nx], ug [B][N][nu], p [B][N+1][5]        inputs
 *   x_out [B][N+1][nx], u_out [B][N][nu], status [B], qp_iter [B]    outputs (qp_iter may be NULL)
 * on_device != 0: all pointers are device pointers (resident in HBM); the call only enqueues work on the handle's
 * stream (use smpc_sync to wait).  on_device == 0: host pointers; the call copies in, runs, copies out and waits. */
int smpc_solve_batch(smpc_handle* h, int B, const double* x0, const double* xg, const double* ug, const double* p,
                     double* x_out, double* u_out, int32_t* status, int32_t* qp_iter, int on_device);

/* Linearisation only (HOT LOOP A of SURVEY 3.2): evaluates every node of every instance at (xg, ug, p) and writes
 * out[B][N+1] records.  Used by the parity tests; pointers follow on_device like smpc_solve_batch. */
int smpc_eval_nodes(smpc_handle* h, int B, const double* xg, const double* ug, const double* p, smpc_node_eval* out,
                    int on_device);

/* ---- callers on either side of the solve ------------------------------------------- */
/* guessCorrection (controller.py:226-231): x_guess[k+1] = f(x_guess[k], u_guess[k]) in place. */
int smpc_guess_correction(smpc_handle* h, int B, double* xg, const double* ug, int on_device);

/* provideControl (controller.py:169-184): per instance, take (x_temp,u_temp) if accept[b] != 0 else keep the old
 * guess; write u_apply = row 0; shift by one and duplicate the last row. */
int smpc_provide_control(smpc_handle* h, int B, const int32_t* accept, const double* x_temp, const double* u_temp,
                         double* xg, double* ug, double* u_apply, int on_device);

/* checkStateConstraints over a trajectory (env_model.py:170-173, 236-243): ok[b] = all nodes within
 * [x_min - tol_x, x_max + tol_x] (the margin-widened model bounds passed here) and collision rows within
 * [lb_chk, ub_chk]; nn_ok[b][k] = g(x_k, alpha) >= -tol_safe (safe_set.py:61-68) if nn_ok != NULL. */
int smpc_check_trajectory(smpc_handle* h, int B, int n_nodes, const double* x, const double* x_min,
                          const double* x_max, double tol_x, const double* row_lb_chk, const double* row_ub_chk,
                          double alpha, double tol_safe, int32_t* state_ok, int32_t* nn_ok, int on_device);

/* plant step AdamModel.integrate (env_model.py:192-206): tau = RNEA_noisy(x,u) + noise, clip, qdd = M^-1(tau - h),
 * double-integrator step.  joints_noisy is [B][nq] smpc_joint (per-instance perturbed inertials) or NULL for the
 * nominal model; tau_noise [B][nq] additive torque noise or NULL. */
int smpc_plant_step(smpc_handle* h, int B, const double* x, const double* u, const smpc_joint* joints_noisy,
                    const double* tau_noise, double* x_next, double* u_eff, int on_device);

/* Closed-loop rollout of the plain RTI policy for n_steps, without a host round trip per step: NaiveController.step
 * (controller.py:274-284: guessCorrection, solve, fails = status == 0 ? 0 : fails + 1, provideControl(fails == 0)) followed
 * by the plant step of scripts/mpc.py:151,240 (smpc_plant_step semantics, optional per-instance model and per-step torque
 * noise).  x_guess / u_guess are the warm start on entry and the shifted guess of the last step on return.
 * Trajectories are STEP-major: x_traj[n_steps+1][B][nx] (x_traj[0] = x0), u_traj[n_steps][B][nu], status_traj[n_steps][B],
 * iter_traj[n_steps][B] (may be NULL), tau_noise[n_steps][B][nq] or NULL, joints_noisy[B][nq] or NULL.
 * Large batches are split into two (B >= 1024) or three (B >= 3072) sub-batches that advance on their own streams inside the
 * engine (worker handles sharing the network weights): the long tail of one sub-batch's QP launch overlaps the bulk of the
 * others'.  The
 * environment variable SMPC_ROLLOUT_STREAMS overrides the number of sub-batches; results do not depend on it. */
int smpc_rollout_batch(smpc_handle* h, int B, int n_steps, const double* x0, double* x_guess, double* u_guess,
                       const double* p, const smpc_joint* joints_noisy, const double* tau_noise, double* x_traj,
                       double* u_traj, int32_t* status_traj, int32_t* iter_traj, int on_device);

/* ---- the policy layer with all state in HBM ------------------------------------------------- */
/* The reference walks its instances one at a time through <Controller>.step (controller.py:274-284, 375-388, 448-498,
 * 524-565, 651-661) and the safe-abort loop of scripts/mpc.py:125-264.  These three entry points are those two pieces of code
 * for B instances at once, state resident on the device, enqueue-only on the handle's stream (no host synchronisation, so a
 * caller can capture a whole closed-loop step in a hipGraph).  All pointers inside the structs and all array arguments are
 * DEVICE pointers unless stated otherwise; flags are one byte per instance (0 / 1). */
enum {
    SMPC_POLICY_NAIVE = 0,          /* NaiveController / TerminalZeroVelocity / STController                      :274-284 */
    SMPC_POLICY_STATE_CHECK = 1,    /* ControllerSafeSetEverywhere: success also needs checkStateConstraints       :651-661 */
    SMPC_POLICY_STWA = 2,           /* STWAController / HTWAController: viable state, abort after N - 1 failures   :375-388 */
    SMPC_POLICY_RECEDING = 3,       /* RecedingController: receding index r, row switched on at node r             :448-498 */
    SMPC_POLICY_REAL_RECEDING = 4   /* RealReceding: node r boxed to the planned state +- tube                     :524-565 */
};

typedef struct {
    int32_t kind;                   /* SMPC_POLICY_* */
    int32_t abort_flag;             /* params.abort_flag (controller.py:471-478) */
    int32_t collision_first_node;   /* != 0: trajectories are collision-tested at their first node only, as the reference's
                                       checkCollision does (env_model.py:238-243); 0: at every node */
    int32_t reserved0;
    double tol_x, alpha, tol_safe;  /* checkStateConstraints / checkSafeConstraints tolerances (env_model.py:170, safe_set.py:61) */
    double tube;                    /* RealReceding: half-width of the box at node r (1e-3, controller.py:531-532) */
    const double *x_min, *x_max;            /* HOST [nx]: model bounds of the state test */
    const double *row_lb_chk, *row_ub_chk;  /* HOST [n_rows]: check bounds of the collision rows */
    const double *stage_lo, *stage_hi;      /* DEVICE [N+1][nx]: RealReceding's bounds away from node r (NULL otherwise) */
} smpc_policy_params;

typedef struct {                    /* what a controller object holds per instance (controller.py:112-131) */
    double *x_guess, *u_guess;      /* [B][N+1][nx], [B][N][nu] */
    double *x_temp, *u_temp;        /* the iterate of the last solve */
    double *p;                      /* [B][N+1][5] */
    double *x_viable;               /* [B][nx] */
    int64_t *fails, *current_step;  /* [B] */
    int64_t *r;                     /* [B] receding index (NULL for the policies without one) */
    int32_t *status, *qp_iter;      /* [B] of the last solve */
    const double* traj;             /* [3][traj_len] reference trajectory of the cost (cost.traj, cost_definition.py:30-31,89), or NULL.
                                       Not NULL: before the solve, p[b][i][0:3] = traj[:, current_step[b] + i] for every node i of the
                                       stepping instances -- what solve() does through ocp_solver.set(i, 'p', .) at
                                       controller.py:153-156 (column index clamped to traj_len - 1).  NULL: p[:, :, 0:3] is left as the
                                       caller set it (the constant ee_ref of the ReachTarget costs). */
    int64_t traj_len;
} smpc_policy_state;

/* <Controller>.step(x) for the instances with stepping[b] != 0 (NULL: all): guessCorrection, the policy's flags / bounds,
 * the RTI solve, the acceptance tests, the fails / r / viable-state automaton, provideControl.  Instances that do not step are
 * left untouched and skipped by the QP kernels.  u_out[b] = the policy's control, u_guess[b][0] for an instance that raises
 * abort, u_other[b] for one that did not step (u_other may be NULL when stepping is).  abort_out[b] = the step's second return
 * value; *any_abort (one int32) is set to 1 if any instance aborted, 0 otherwise. */
int smpc_policy_step(smpc_handle* h, int B, const smpc_policy_params* par, const smpc_policy_state* st, const double* x,
                     const uint8_t* stepping, const double* u_other, double* u_out, uint8_t* abort_out, int32_t* any_abort);

typedef struct {                    /* the driver's per-instance state (scripts/mpc.py:102-124) */
    double* x_cur;                  /* [B][nx] */
    uint8_t *alive, *sa, *collided; /* [B]: still simulated / following a backup trajectory / failed */
    int64_t *ja, *last_x, *last_u;  /* [B]: abort clock; last valid row of the state / input logs */
    double *x_abort, *u_abort;      /* [B][Nb+1][nx], [B][Nb][nu]: backup trajectories */
    int64_t* step;                  /* [1]: the step counter j */
    double *x_log, *u_log;          /* step-major logs [n_steps+1][B][nx], [n_steps][B][nu] */
    int64_t* r_log;                 /* [n_steps][B] receding index used at each step, -1 where none (or NULL) */
    uint8_t* resumed;               /* [B]: written by smpc_loop_pre -- the instance left its backup trajectory at THIS step and
                                       steps its controller again (scripts/mpc.py:137-141); may be NULL */
} smpc_loop_state;

/* scripts/mpc.py:130-151 before the controller's step: PD tracking of the backup trajectory / hold / resume for the instances
 * in safe abort -> u_other[B][nu]; stepping[b] = alive and not in abort; logs r (may be NULL) of the stepping instances. */
int smpc_loop_pre(smpc_handle* h, int B, int Nb, const smpc_loop_state* ls, const int64_t* r, const uint8_t* pending,
                  double* u_other, uint8_t* stepping);

/* What the driver makes of the aborts the controllers raised in this step (abort[B] = smpc_policy_step's abort_out, updated
 * in place).  An abort raised by an instance that stepped normally is an abort EVENT (scripts/mpc.py:161-190: viable state
 * recorded, backup OCP solved): it stays set.  An abort raised on the very step an instance resumed MPC after a backup
 * trajectory (ls->resumed) happens inside the `if sa_flag:` branch of the reference (mpc.py:137-141), where no event is
 * opened: the instance simply is in safe abort again -- old backup trajectory, abort clock still running -- and re-tests its
 * velocity next step.  With reference_quirks != 0 that is what happens here (sa[b] = 1, abort[b] = 0); with 0 every abort is an
 * event.  *any_event (one int32) = 1 if an event remains, else 0. */
int smpc_loop_classify_aborts(smpc_handle* h, int B, const smpc_loop_state* ls, int reference_quirks, uint8_t* abort,
                              int32_t* any_event);

/* scripts/mpc.py:161-190, second half: the n_c abort events of the previous step, applied once their backup OCPs (solved as a
 * compact batch, possibly on another handle / stream that the caller has ordered before this call) are known.  rows[n_c] =
 * instance of each event, status_c[n_c] / x_c[n_c][Nb+1][nx] / u_c[n_c][Nb][nu] = the backup solves.  Solved: the instance
 * follows its backup trajectory from this step on (u[b] = PD law on its first node, clock 1), viable[b] += 1 (a saturating
 * count: mpc.py:189 appends the instance to viable_idx once per event and :277-278 removes it once); failed: lost at
 * the step of the event.  pending[b] (the flag smpc_loop_pre reads) is cleared.  Between the event and this call an instance
 * only has to be kept from stepping, which is what lets the backup solve overlap the next step's solve. */
int smpc_loop_apply_backup(smpc_handle* h, int B, int Nb, const smpc_loop_state* ls, int n_c, const int64_t* rows,
                           const int32_t* status_c, const double* x_c, const double* u_c, uint8_t* viable, double* u,
                           uint8_t* pending);

/* scripts/mpc.py:240-264 after it: logs u, plant step (smpc_plant_step semantics), state test of the new state
 * (par: x_min / x_max / tol_x / row check bounds), logs, outcome flags, next current state, j += 1. */
int smpc_loop_post(smpc_handle* h, int B, const smpc_policy_params* par, const smpc_loop_state* ls, const double* u,
                   const smpc_joint* joints_noisy, const double* tau_noise);

/* wait for the handle's stream */
int smpc_sync(smpc_handle* h);
/* the hipStream_t the handle enqueues on (for event timing by the caller) */
void* smpc_stream(smpc_handle* h);
/* device time of the kernels of the last smpc_solve_batch, measured with HIP events on the handle's stream:
 * ms[0] linearise (since round 4: the stage builder -- linearisation AND the set-up of the QP's stage records, one kernel),
 * ms[1] MLP, ms[2] QP (the interior point; with SMPC_STAGE_BUILD=0 also k_qp_setup), ms[3] total.
 * Mirrors ocp_solver.get_stats('time_lin'|'time_qp'|'time_tot')
 * (controller.py:123-124,192-193).  Only valid when timing was enabled.  on = 1: HIP events + the in-kernel load-balance probe of
 * smpc_get_qp_wave_stats; on = 2: HIP events only (what a running loop can afford); 0: off. */
int smpc_enable_timing(smpc_handle* h, int on);
int smpc_get_timing(smpc_handle* h, float* ms4);
/* Split of ms[2] of smpc_get_timing: ms2[0] = k_qp_setup (stage records + initial point; ~0 on the default path, where the stage
 * builder has already written them), ms2[1] = k_qp_ipm (the interior-point iterations) -- the per-kernel durations rocprofv3
 * --kernel-trace reports (acados: time_qp_solver_call). */
int smpc_get_qp_timing(smpc_handle* h, float* ms2);
/* The same durations for the solve `back` solves before the last one (0 = the last; the handle keeps the events of its last 64
 * timed solves), without waiting: ms6 = {linearise, MLP, k_qp_setup, k_qp_ipm, total, valid}.  valid = 0 (and the rest 0) when
 * there is no such solve or it has not finished yet.  Lets a loop that enqueues far ahead of the GPU collect per-kernel times of
 * its own launches afterwards (bench.py: kernel_ms_in_loop). */
int smpc_get_timing_history(smpc_handle* h, int back, float* ms6);
/* acc3[0] += sum of qp_iter[B], acc3[1] += number of status[b] != 0, acc3[2] += B -- device-side counters of a closed loop
 * (DEVICE pointers; qp_iter may be NULL), enqueued on the handle's stream. */
int smpc_accumulate_stats(smpc_handle* h, int B, const int32_t* status, const int32_t* qp_iter, unsigned long long* acc3);
/* Load balance of the last timed k_qp_ipm launch: out3[0] = mean busy time of a half-wavefront (= one instance), out3[1] =
 * first start to last end of any half-wavefront, both in microseconds of the constant 100 MHz clock, out3[2] = half-waves
 * counted.  out3[1] / out3[0] is the share of the launch spent waiting for its slowest instances. */
int smpc_get_qp_wave_stats(smpc_handle* h, double* out3);

#ifdef __cplusplus
}
#endif
#endif /* SMPC_H_ */
