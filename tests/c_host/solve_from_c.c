/* A plain C host of liblpx.so: what a maintainer's JNI / cgo glue ends up calling, with no Python and no torch in the
 * process.  Solves the reference's Spock LPs (LPSolverSpec.groovy:76-111) on one device and — with the same row
 * blocks on two "devices" (both ordinal 0 when only one GPU exists: LPX_HOST_DEVICES="0,0") — through lpx_solve_multi,
 * then a dense random LP for a fixed pivot budget on both paths, and prints one line per result for the test to
 * compare.  cc -I include solve_from_c.c -L<libdir> -llpx */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lpx.h"

static int parse_devices(int32_t* dev) {
  const char* e = getenv("LPX_HOST_DEVICES");
  int n = 0;
  if (!e || !*e) e = "0,0";
  char buf[128];
  strncpy(buf, e, sizeof buf - 1);
  buf[sizeof buf - 1] = 0;
  for (char* t = strtok(buf, ","); t && n < LPX_MAX_DEVICES; t = strtok(NULL, ",")) dev[n++] = atoi(t);
  return n;
}

static void report(const char* name, int status, const lpx_solve_result* r) {
  printf("%s status %d text %s p1 %lld p2 %lld x0 %d objbits %016llx msg \"%s\"\n", name, status, r->objective_text,
         (long long)r->pivots_phase1, (long long)r->pivots_phase2, r->x0_slot,
         (unsigned long long)*(const unsigned long long*)&r->objective, lpx_status_message(status));
}

int main(void) {
  int32_t dev[LPX_MAX_DEVICES];
  const int ndev = parse_devices(dev);
  if (lpx_device_count() < 1) { fprintf(stderr, "no HIP device\n"); return 2; }
  lpx_solve_result r;
  /* LPSolverSpec.groovy:76-87: max x + y, 4x - y <= 8, 2x + y <= 10, -5x + 2y <= 2 -> 8 */
  {
    const double A[] = {4, -1, 2, 1, -5, 2}, b[] = {8, 10, 2}, c[] = {1, 1};
    int st = lpx_solve(3, 2, A, 2, b, c, 1, NULL, &r);
    report("spec8 one", st, &r);
    st = lpx_solve_multi(3, 2, A, 2, b, c, 1, NULL, dev, ndev < 3 ? ndev : 3, &r);
    report("spec8 multi", st, &r);
  }
  /* LPSolverSpec.groovy:100-111: infeasible start (phase 1) -> 20 */
  {
    const double A[] = {1, 0, -1, 0, 0, 1, 0, -1}, b[] = {10, -2, 10, -2}, c[] = {1, 1};
    int st = lpx_solve(4, 2, A, 2, b, c, 1, NULL, &r);
    report("phase1 one", st, &r);
    st = lpx_solve_multi(4, 2, A, 2, b, c, 1, NULL, dev, ndev, &r);
    report("phase1 multi", st, &r);
  }
  /* infeasible: x + y <= 1, x + y >= 3 — perm_out exactly n + m entries, guard behind it */
  {
    const double A[] = {1, 1, -1, -1}, b[] = {1, -3}, c[] = {1, 1};
    int32_t perm[4 + 2] = {7, 7, 7, 7, 12345, 12345};
    lpx_solve_options o;
    memset(&o, 0, sizeof o);
    o.max_pivots = -1;
    o.perm_out = perm;
    int st = lpx_solve(2, 2, A, 2, b, c, 1, &o, &r);
    report("infeasible one", st, &r);
    st = lpx_solve_multi(2, 2, A, 2, b, c, 1, &o, dev, ndev < 2 ? ndev : 2, &r);
    report("infeasible multi", st, &r);
    printf("guard %d %d\n", perm[4], perm[5]);
  }
  /* dense random LP, LPState level: 150 pivots on one device and through the multi handle; checksums must agree */
  {
    const int m = 512, n = 1024;
    double *A = malloc(sizeof(double) * m * n), *b = malloc(sizeof(double) * m), *c = malloc(sizeof(double) * n);
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < m * n + m + n; i++) {
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      const double u = (double)(s >> 11) / 9007199254740992.0;
      if (i < m * n) A[i] = u; else if (i < m * n + m) b[i - m * n] = (n / 4.0) * (1.0 + u); else c[i - m * n - m] = u;
    }
    lpx_state* one = NULL;
    lpx_multi* many = NULL;
    int rc = lpx_state_create(m, n, A, n, b, c, 0.0, NULL, 0, m, dev[0], &one);
    if (!rc) rc = lpx_multi_create(m, n, A, n, b, c, 0.0, NULL, dev, ndev, &many);
    int64_t p1 = 0, p2 = 0;
    int32_t s1 = 0, s2 = 0;
    uint64_t h1[3] = {0, 0, 0}, h2[3] = {0, 0, 0};
    double v1 = 0, v2 = 0;
    if (!rc) rc = lpx_simplex_loop(one, 150, &p1, &s1, NULL);
    if (!rc) rc = lpx_multi_simplex_loop(many, 150, &p2, &s2, NULL);
    if (!rc) rc = lpx_state_checksum(one, h1);
    if (!rc) rc = lpx_multi_checksum(many, h2);
    if (!rc) rc = lpx_state_read(one, NULL, n, NULL, NULL, &v1, NULL);
    if (!rc) rc = lpx_multi_read(many, NULL, n, NULL, NULL, &v2, NULL);
    printf("dense rc %d one %lld/%d %016llx %016llx %016llx v %016llx\n", rc, (long long)p1, s1, (unsigned long long)h1[0],
           (unsigned long long)h1[1], (unsigned long long)h1[2], *(unsigned long long*)&v1);
    printf("dense rc %d multi %lld/%d %016llx %016llx %016llx v %016llx\n", rc, (long long)p2, s2, (unsigned long long)h2[0],
           (unsigned long long)h2[1], (unsigned long long)h2[2], *(unsigned long long*)&v2);
    if (rc) printf("error: %s\n", lpx_last_error());
    lpx_state_destroy(one);
    lpx_multi_destroy(many);
    free(A); free(b); free(c);
  }
  return 0;
}
