// Host build of the product's closed-form inverse-dynamics derivatives (safe_mpc_amd/csrc/rnea_deriv.hpp: the same source the
// HIP kernels compile), exposed with C linkage so that tests/test_rnea_deriv.py can compare it with the oracle's dual numbers
// without a GPU.
#include "../../safe_mpc_amd/csrc/rnea_deriv.hpp"

extern "C" int host_rnea_with_derivatives(int nq, const smpc_joint* J, const double* grav, const double* q, const double* qd,
                                          const double* qdd, double* tau, double* M, double* dq, double* dv) {
    switch (nq) {
    case 5: smpc::rd::rnea_with_derivatives<5>(J, grav, q, qd, qdd, tau, M, dq, dv); return 0;
    case 6: smpc::rd::rnea_with_derivatives<6>(J, grav, q, qd, qdd, tau, M, dq, dv); return 0;
    case 7: smpc::rd::rnea_with_derivatives<7>(J, grav, q, qd, qdd, tau, M, dq, dv); return 0;
    }
    return -1;
}
