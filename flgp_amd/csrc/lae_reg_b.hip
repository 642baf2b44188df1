// Register-resident LAE, r = 2..9 -- see lae_reg.h.
#include "lae_reg.h"

namespace flgp {
int launch_lae_reg_lo(FLGP_LAE_REG_ARGS, int r, int force_dpl, int force_lp) {
  switch (r) {
    case 2: return launch_lae_reg_r<2>(FLGP_LAE_REG_PASS, force_dpl, force_lp);
    case 3: return launch_lae_reg_r<3>(FLGP_LAE_REG_PASS, force_dpl, force_lp);
    case 4: return launch_lae_reg_r<4>(FLGP_LAE_REG_PASS, force_dpl, force_lp);
    case 5: return launch_lae_reg_r<5>(FLGP_LAE_REG_PASS, force_dpl, force_lp);
    case 6: return launch_lae_reg_r<6>(FLGP_LAE_REG_PASS, force_dpl, force_lp);
    case 7: return launch_lae_reg_r<7>(FLGP_LAE_REG_PASS, force_dpl, force_lp);
    case 8: return launch_lae_reg_r<8>(FLGP_LAE_REG_PASS, force_dpl, force_lp);
    case 9: return launch_lae_reg_r<9>(FLGP_LAE_REG_PASS, force_dpl, force_lp);
  }
  return FLGP_LAE_REG_NONE;
}
}  // namespace flgp
