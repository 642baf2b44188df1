// Register-resident LAE, r = 10 (the headline shape) and the dispatcher -- see lae_reg.h.
#include "lae_reg.h"

namespace flgp {
int launch_lae_reg_lo(FLGP_LAE_REG_ARGS, int r, int force_dpl, int force_lp);
int launch_lae_reg_hi(FLGP_LAE_REG_ARGS, int r, int force_dpl, int force_lp);

int launch_lae_reg(hipStream_t st, const double *dX, int n, int ldx, int d, const double *dUt, int dpad, int r,
                   const int *d_knn, int ldk, int *d_ei, double *d_ev) {
  const int fd = tuning("lae_dpl", 0), fl = tuning("lae_lp", 0);
  if (r == 10) return launch_lae_reg_r<10>(FLGP_LAE_REG_PASS, fd, fl);
  if (r >= 2 && r < 10) return launch_lae_reg_lo(FLGP_LAE_REG_PASS, r, fd, fl);
  if (r > 10 && r <= 16) return launch_lae_reg_hi(FLGP_LAE_REG_PASS, r, fd, fl);
  return FLGP_LAE_REG_NONE;
}
}  // namespace flgp
