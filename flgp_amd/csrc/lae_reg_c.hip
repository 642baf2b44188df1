// Register-resident LAE, r = 11..16 (d <= 32) -- see lae_reg.h.
#include "lae_reg.h"

namespace flgp {
int launch_lae_reg_hi(FLGP_LAE_REG_ARGS, int r, int force_dpl, int force_lp) {
  switch (r) {
    case 11: return launch_lae_reg_r<11>(FLGP_LAE_REG_PASS, force_dpl, force_lp);
    case 12: return launch_lae_reg_r<12>(FLGP_LAE_REG_PASS, force_dpl, force_lp);
    case 13: return launch_lae_reg_r<13>(FLGP_LAE_REG_PASS, force_dpl, force_lp);
    case 14: return launch_lae_reg_r<14>(FLGP_LAE_REG_PASS, force_dpl, force_lp);
    case 15: return launch_lae_reg_r<15>(FLGP_LAE_REG_PASS, force_dpl, force_lp);
    case 16: return launch_lae_reg_r<16>(FLGP_LAE_REG_PASS, force_dpl, force_lp);
  }
  return FLGP_LAE_REG_NONE;
}
}  // namespace flgp
