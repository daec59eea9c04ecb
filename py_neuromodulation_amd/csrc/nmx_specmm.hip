// nmx_specmm.hip -- translation unit of the matrix-pipe spectrum kernel (nmx_k_specmm.h): FFT band power + Hjorth /
// LineLength / Raw of 1000-sample windows; 16 windows per wave, one persistent four-wave workgroup per CU (the waves share
// nothing: the DFT table lives in each lane's registers, the LDS is four rings of step buffers).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "nmx_k_specmm.h"

extern __shared__ __attribute__((aligned(16))) char nmx_smem_smm[];

// one wave per SIMD: the 512-entry register file holds a lane's 256 table entries next to its working set
template <int NB, bool TD, bool CLEAN>
__global__ void __launch_bounds__(256, 1) nmx_kern_specmm_w1000(const NmxTimeOscArgs A0, long long n_items) {
  typedef const NmxTimeOscArgs __attribute__((address_space(4)))* nmx_karg_p;
  nmx_karg_p Ap = (nmx_karg_p)__builtin_amdgcn_kernarg_segment_ptr();
  const NmxTimeOscArgs& A = *(const NmxTimeOscArgs*)Ap;
  const int tid = (int)threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const unsigned lds = nmx_lds_addr(nmx_smem_smm);
  const int C = A.n_channels, n_windows = (int)(n_items / C);
  const long long n_tiles = (long long)((n_windows + 15) / 16) * C, stride = (long long)gridDim.x * NMX_SMM_WAVES;
  long long t = (long long)blockIdx.x * NMX_SMM_WAVES + wave;
  if (t >= n_tiles) return;
  typedef NmxSmmWave<NB, TD, CLEAN> Wave;
  constexpr int R = Wave::R;
  Wave W(Ap, lds, wave, lane);
  W.rows(t, n_windows);
  W.adopt();
  // the first R steps of the first tile, one per slot; then step 0 into registers
  W.template dma<0>(0, W.src0, W.src1);
  W.template dma<1>(1, W.src0, W.src1);
  W.template dma<2>(2, W.src0, W.src1);
  if (R > 3) W.template dma<3>(3, W.src0, W.src1);
  if (R > 4) W.template dma<4>(4, W.src0, W.src1);
  nmx_smm_wait_vm<8 * (R - 1)>();
  NmxSmmRegs Cu;
  W.template read<0>(0, 0, Cu);
#pragma unroll 1
  for (;;) {
    const long long tn = t + stride;
    const bool more = tn < n_tiles;
    typename Wave::Tile T;
    Wave::clear(T);
    W.template step<0>(T, Cu, more, tn, n_windows);
    W.template step<1>(T, Cu, more, tn, n_windows);
    W.template step<2>(T, Cu, more, tn, n_windows);
    W.template step<3>(T, Cu, more, tn, n_windows);
    W.template step<4>(T, Cu, more, tn, n_windows);
    W.template step<5>(T, Cu, more, tn, n_windows);
    W.template step<6>(T, Cu, more, tn, n_windows);
    W.template step<7>(T, Cu, more, tn, n_windows);
    W.finish(T, t, n_windows);
    if (!more) break;
    W.adopt();
    t = tn;
  }
  W.flush();
}

extern "C" void nmx_wave_launch_timeosc_w1000_todo(const NmxTimeOscArgs* A, int n_items, hipStream_t s);

// returns 0 when the configuration needs another kernel, 1 / 2 when launched (2: windows may have been flagged)
extern "C" int nmx_specmm_launch(const NmxTimeOscArgs* A, int n_items, hipStream_t s) {
  static int on = -1;
  if (on < 0) { const char* v = getenv("NMX_SPECMM"); on = (v && v[0] == '0') ? 0 : 1; }
  if (!on || !nmx_specmm_ok(*A) || n_items < 1) return 0;
  static thread_local int n_cu = 0, n_cu_dev = -1;   // per host thread and device (multi-device streams)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  if (!n_cu || n_cu_dev != dev) {
    hipDeviceProp_t prop;
    n_cu = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.multiProcessorCount : 256;
    n_cu_dev = dev;
  }
  const int n_win = n_items / A->n_channels;
  const long long n_tiles = (long long)((n_win + 15) / 16) * A->n_channels;
  long long grid = (n_tiles + NMX_SMM_WAVES - 1) / NMX_SMM_WAVES;
  if (grid > n_cu) grid = n_cu;
  const size_t lds = (size_t)NMX_SMM_LDS_BYTES;
  const bool td = (A->features & (NMXD_F_HJORTH | NMXD_F_LINELENGTH | NMXD_F_RAW)) != 0, clean = false;   // (cleaning is the todo kernel's: nmx_k_specmm.h, finish())
#define NMX_SMM_LAUNCH(NB, TD, CL)                                                                                       \
  do {                                                                                                                   \
    static unsigned long long seen = 0; /* per instantiation and device */                                               \
    if (nmx_first_on_device(seen)) (void)hipFuncSetAttribute((const void*)nmx_kern_specmm_w1000<NB, TD, CL>,                                 \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, NMX_SMM_LDS_BYTES);                 \
    hipLaunchKernelGGL((nmx_kern_specmm_w1000<NB, TD, CL>), dim3((unsigned)grid), dim3(256), lds, s, *A, (long long)n_items); \
  } while (0)
#ifdef NMX_SMM_SINGLE
  (void)td; (void)clean;
  NMX_SMM_LAUNCH(4, true, false);
#else
  if (A->n_bands <= 4) {
    if (td && clean) NMX_SMM_LAUNCH(4, true, true); else if (td) NMX_SMM_LAUNCH(4, true, false);
    else if (clean) NMX_SMM_LAUNCH(4, false, true); else NMX_SMM_LAUNCH(4, false, false);
    nmxi_note_kernel("nmx_kern_specmm_w1000<4>");
  } else {
    if (td && clean) NMX_SMM_LAUNCH(8, true, true); else if (td) NMX_SMM_LAUNCH(8, true, false);
    else if (clean) NMX_SMM_LAUNCH(8, false, true); else NMX_SMM_LAUNCH(8, false, false);
    nmxi_note_kernel("nmx_kern_specmm_w1000<8>");
  }
#endif
#undef NMX_SMM_LAUNCH
  // 2: the caller launches the pass over the flagged windows (nmx_wave_launch_timeosc_w1000_todo) -- behind its stage
  // timer, which brackets THIS kernel alone (bench.py: roofline_modeA is this kernel's launch duration)
  return 2;   // (a window's power can overflow behind a cleaning stage too: samples on the rail, re-referenced)
}
