/* abi_smoke.c -- the C ABI used from plain C (no Python, no C++): version, device count, error
 * reporting and argument validation of include/nmx.h.  Built and run by tests/test_abi.py with
 *   gcc abi_smoke.c -I include -L py_neuromodulation_amd -lnmx -Wl,-rpath,...
 * It makes no compute call (the CPU test tier has no GPU); on a GPU box it additionally creates and
 * destroys a minimal plan (one channel, Raw feature). */
#include <stdio.h>
#include <string.h>

#include "nmx.h"

int main(void) {
  if (nmx_abi_version() != NMX_ABI_VERSION) { printf("FAIL abi version\n"); return 1; }
  const int ndev = nmx_device_count();
  if (ndev < 0) { printf("FAIL device count\n"); return 1; }
  nmx_plan* plan = NULL;
  if (nmx_plan_create(NULL, &plan) >= 0) { printf("FAIL null desc accepted\n"); return 1; }
  if (!nmx_last_error() || !strlen(nmx_last_error())) { printf("FAIL no error text\n"); return 1; }

  nmx_plan_desc d;
  memset(&d, 0, sizeof d);
  d.abi_version = NMX_ABI_VERSION;
  d.n_channels = 1;
  d.window = 2;                 /* too short: must be rejected with NMX_E_INVALID before any device work */
  d.sfreq = 1000.0;
  d.feat_hz = 10.0;
  if (nmx_plan_create(&d, &plan) != NMX_E_INVALID) { printf("FAIL short window accepted\n"); return 1; }

  nmx_norm* norm = NULL;
  if (nmx_norm_create(0, 0, NMX_NORM_ZSCORE, 3.0f, 300, NULL, &norm) != NMX_E_INVALID) {
    printf("FAIL bad normaliser accepted\n");
    return 1;
  }
  if (ndev > 0) {               /* GPU box: a real (tiny) plan */
    d.window = 64;
    d.features = NMX_F_RAW;
    d.n_outputs = 1;
    d.raw_cols.base = 0; d.raw_cols.ch_stride = 1;
    if (nmx_plan_create(&d, &plan) != 0) { printf("FAIL plan: %s\n", nmx_last_error()); return 1; }
    double x[64];
    for (int i = 0; i < 64; ++i) x[i] = i;
    float out = -1.f;
    if (nmx_process_window(plan, x, 64, &out, NULL) != 0 || out != 63.f) {
      printf("FAIL raw feature %g: %s\n", (double)out, nmx_last_error());
      return 1;
    }
    nmx_plan_destroy(plan);
  }
  printf("OK devices=%d\n", ndev);
  return 0;
}
