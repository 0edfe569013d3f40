/* Plain-C consumer of libnbmf_hip.so: proves the boundary is a C ABI (no Python, no C++ types).
 * Build: gcc -std=c99 -I include tests/c/abi_smoke.c -L nbmf_mm_amd -lnbmf_hip -Wl,-rpath,$PWD/nbmf_mm_amd -lm -o build/abi_smoke
 * Fits a 96 x 80 binary matrix with K = 5 for 40 iterations and checks the invariants the reference's
 * own tests check (monotone loss, simplex columns, H in range); exit code 0 on success. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "nbmf_hip.h"

#define M 96
#define N 80
#define K 5
#define ITERS 40

static unsigned long long s = 88172645463325252ull;
static double urand(void) { /* xorshift64 */
  s ^= s << 13; s ^= s >> 7; s ^= s << 17;
  return (double)(s >> 11) / 9007199254740992.0;
}

#define CK(call) do { int rc_ = (call); if (rc_ != NBMF_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, nbmf_last_error()); return 1; } } while (0)

int main(void) {
  static double Y[M * N], mask[M * N], W[K * M], H[K * N], losses[ITERS];
  int i, j, k, n_iter = 0, flags = 0, ndev = 0;
  double n_obs = 0, loss_now = 0, ll = 0;
  nbmf_ctx* ctx = NULL;
  if (nbmf_abi_version() != 4) return 2;
  if (nbmf_device_count(&ndev) != NBMF_OK || ndev < 1) { fprintf(stderr, "no GPU: %s\n", nbmf_last_error()); return 3; }
  for (i = 0; i < M * N; ++i) { Y[i] = urand() < 0.3 ? 1.0 : 0.0; mask[i] = urand() < 0.9 ? 1.0 : 0.0; }
  for (i = 0; i < M; ++i) {           /* W: k x m, columns on the simplex (_solver.py:132-136) */
    double sum = 0;
    for (k = 0; k < K; ++k) { W[k * M + i] = 0.1 + 0.8 * urand(); sum += W[k * M + i]; }
    for (k = 0; k < K; ++k) W[k * M + i] /= sum;
  }
  for (i = 0; i < K * N; ++i) H[i] = 0.1 + 0.8 * urand();
  CK(nbmf_create(M, N, K, 0, &ctx));
  CK(nbmf_set_hyper(ctx, 1.2, 1.2, 1e-8, NBMF_PROJ_NORMALIZE));
  CK(nbmf_upload(ctx, Y, N, 0, mask, NBMF_MASK_F64, N, &flags));
  if (!(flags & NBMF_FLAG_BINARY_PATH)) { fprintf(stderr, "expected the 1-byte storage path\n"); return 4; }
  CK(nbmf_get_n_obs(ctx, &n_obs));
  CK(nbmf_set_factors(ctx, W, H));
  CK(nbmf_run(ctx, ITERS, 0.0, losses, &n_iter));
  CK(nbmf_get_factors(ctx, W, H));
  CK(nbmf_loss(ctx, &loss_now));
  CK(nbmf_loglik(ctx, 0, &ll));
  CK(nbmf_destroy(ctx));
  if (n_iter != ITERS) return 5;
  for (i = 1; i < ITERS; ++i) if (!(losses[i] <= losses[i - 1] + 1e-12)) { fprintf(stderr, "loss not monotone at %d\n", i); return 6; }
  /* (a small problem runs inside one persistent kernel, nbmf_loss is a sweep of its own: same value to rounding) */
  if (!(fabs(loss_now - losses[ITERS - 1]) <= 1e-13 * fabs(loss_now))) { fprintf(stderr, "nbmf_loss disagrees with the last loss of nbmf_run\n"); return 7; }
  for (i = 0; i < M; ++i) { double sum = 0; for (k = 0; k < K; ++k) sum += W[k * M + i]; if (fabs(sum - 1.0) > 1e-12) return 8; }
  for (j = 0; j < K * N; ++j) if (!(H[j] >= 1e-8 && H[j] <= 1.0 - 1e-8)) return 9;
  if (!(ll < 0.0) || !(n_obs > 0.5 * M * N)) return 10;
  printf("abi_smoke ok: %d iterations, loss %.15f -> %.15f, n_obs %.0f\n", n_iter, losses[0], losses[ITERS - 1], n_obs);
  return 0;
}
