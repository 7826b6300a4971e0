/* Exhaustive check behind flx_filter.hip: for every byte k, k/255 from one multiply and one FMA correction equals the IEEE
 * division, int(k/255 * 255) == k, and k/255 > 0.1 (or >= 0.1) exactly when k >= 26.   gcc -O0 -ffp-contract=off tools/unorm_check.c -lm */
#include <stdio.h>
#include <math.h>
int main(void) {
  const float r = 1.0f / 255.0f;
  int bad_mul = 0, bad_m1 = 0, bad_idw = 0, thr = -1;
  for (int k = 0; k < 256; k++) {
    const float want = (float)k / 255.0f;
    float q = (float)k * r;
    if (q != want) bad_mul++;
    float e = fmaf(-q, 255.0f, (float)k);
    float q2 = fmaf(e, r, q);
    if (q2 != want) bad_m1++;
    if ((int)(want * 255.0f) != k) bad_idw++;
    if (thr < 0 && want > 0.1f) thr = k;
  }
  printf("k*r mismatches %d, one Markstein step mismatches %d, (int)(k/255*255) != k: %d, first k with k/255 > 0.1f: %d, 0.1f == k/255 for some k: ", bad_mul, bad_m1, bad_idw, thr);
  int eq = 0; for (int k = 0; k < 256; k++) if ((float)k / 255.0f == 0.1f) eq = 1; printf("%d\n", eq);
  return 0;
}
