/* Checks the identity the LAE kernels use for (cumsum - 1) / j (flgp_amd/csrc/lae_dev.h, div_const):
 *   q0 = RN(x * RN(1/J));  rem = fma(-J, q0, x);  q = fma(rem, RN(1/J), q0)  ==  RN(x / J)
 * for every small integer J and x in the range the kernels admit (|x| in [2^-53, 2^900] or 0).
 * Prints the number of mismatches; exit status 0 iff none. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static uint64_t s = 88172645463325252ULL;
static uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
int main(int argc, char **argv) {
  const long per_j = argc > 1 ? atol(argv[1]) : 2000000;
  long bad = 0, tot = 0;
  for (int J = 1; J <= 32; ++J) {
    const double y = (double)J, rj = 1.0 / y;
    for (long it = 0; it < per_j; ++it) {
      uint64_t b = rnd();
      double x;
      switch (it & 3) {
        case 0: { uint64_t e = 1023 - 53 + (rnd() % 110); b = (b & 0x800fffffffffffffULL) | (e << 52); break; }
        case 1: { uint64_t e = 1023 - 53 + (rnd() % 953); b = (b & 0x800fffffffffffffULL) | (e << 52); break; }
        case 2: { double m = (double)(rnd() >> 11); x = y * m; memcpy(&b, &x, 8); b += (uint64_t)((int64_t)(rnd() % 5) - 2); break; }
        default: { double c = (double)(rnd() >> 11) * 0x1p-53 * 3.0; x = c - 1.0; memcpy(&b, &x, 8); break; }
      }
      memcpy(&x, &b, 8);
      if (!(fabs(x) <= 0x1p900)) continue;
      const double q0 = x * rj, rem = fma(-y, q0, x), q = fma(rem, rj, q0), ref = x / y;
      ++tot;
      if (!(q == ref)) { if (bad < 5) printf("J=%d x=%a q=%a ref=%a\n", J, x, q, ref); ++bad; }
    }
  }
  printf("checked %ld mismatches %ld\n", tot, bad);
  return bad != 0;
}
