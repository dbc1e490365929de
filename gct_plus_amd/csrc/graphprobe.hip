// Diagnostics of hipGraph replay on the box this library runs on (tools/graph_probe.py, bench.py's decode block,
// KVDecoder's replay guard).  Round 2 met boxes on which replaying the captured decode step (~70 kernel nodes) was
// 3-13x slower than launching the same kernels one by one; these probes separate the candidate causes with kernels that
// do nothing else: the per-node cost of a replay, the cost of fetching a several-hundred-byte by-value kernel argument
// block from wherever the graph keeps it, and the dynamic-LDS opt-in the large GEMM nodes carry.
#include <string.h>

#include "common.h"
#include "../../include/gctplus_diag.h"

namespace {

struct BigArgs {          // the size of gemm.hip's GemmArgs (320 B)
  int64_t v[40];
};

__global__ void probe_tiny_kernel(int* sink) {
  if (sink && threadIdx.x == 999999) *sink = 1;
}

// every wave reads the whole argument block (scalar loads from the kernarg segment) and keeps it live
__global__ __launch_bounds__(512) void probe_bigargs_kernel(const BigArgs a, int64_t* sink) {
  int64_t s = 0;
#pragma unroll
  for (int i = 0; i < 40; ++i) s += a.v[i];
  if (s == 0x7fffffffffffll) *sink = s;     // never true for the values the host passes
}

__global__ __launch_bounds__(512) void probe_bigargs_lds_kernel(const BigArgs a, int64_t* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  int64_t s = 0;
#pragma unroll
  for (int i = 0; i < 40; ++i) s += a.v[i];
  if (threadIdx.x == 0) lds[0] = (unsigned char)s;
  __syncthreads();
  if (s + lds[0] == 0x7fffffffffffll) *sink = s;
}

// the same argument block behind one pointer into device memory
__global__ __launch_bounds__(512) void probe_ptrargs_kernel(const BigArgs* __restrict__ a, int64_t* sink) {
  int64_t s = 0;
#pragma unroll
  for (int i = 0; i < 40; ++i) s += a->v[i];
  if (s == 0x7fffffffffffll) *sink = s;
}

constexpr int PROBE_LDS = 144 * 1024;

int launch_variant(int variant, hipStream_t st, const BigArgs& a, const BigArgs* dev_args, int64_t* sink) {
  switch (variant) {
    case 0:
      hipLaunchKernelGGL(probe_tiny_kernel, dim3(1), dim3(64), 0, st, (int*)sink);
      break;
    case 1:
      hipLaunchKernelGGL(probe_bigargs_kernel, dim3(2048), dim3(512), 0, st, a, sink);
      break;
    case 2:
      hipLaunchKernelGGL(probe_bigargs_lds_kernel, dim3(2048), dim3(512), PROBE_LDS, st, a, sink);
      break;
    case 3:
      hipLaunchKernelGGL(probe_ptrargs_kernel, dim3(2048), dim3(512), 0, st, dev_args, sink);
      break;
    default:
      gct_set_error("graph_probe: unknown variant %d", variant);
      return GCT_ERR_ARG;
  }
  GCT_LAUNCH_CHECK("graph_probe");
  return GCT_OK;
}

#define HIPC(call)                                                                 \
  do {                                                                             \
    hipError_t e__ = (call);                                                       \
    if (e__ != hipSuccess) {                                                       \
      gct_set_error("graph_probe: %s: %s", #call, hipGetErrorString(e__));         \
      rc = GCT_ERR_HIP;                                                            \
      goto done;                                                                   \
    }                                                                              \
  } while (0)

}  // namespace

extern "C" int gct_graph_probe(int variant, int nodes, int reps, float* eager_ms, float* graph_ms,
                               int32_t* graph_nodes) {
  GCT_CHECK_ARG(variant >= 0 && variant <= 3 && nodes >= 1 && nodes <= 4096 && reps >= 1 && reps <= 1000 && eager_ms &&
                    graph_ms,
                "graph_probe: bad args");
  int rc = GCT_OK;
  hipStream_t st = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipGraph_t graph = nullptr;
  hipGraphExec_t exec = nullptr;
  int64_t* sink = nullptr;
  BigArgs* dev_args = nullptr;
  BigArgs a;
  float ms = 0.f;
  size_t nn = 0;
  for (int i = 0; i < 40; ++i) a.v[i] = i + 1;
  HIPC(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  HIPC(hipEventCreate(&e0));
  HIPC(hipEventCreate(&e1));
  HIPC(hipMalloc(&sink, 64));
  HIPC(hipMalloc(&dev_args, sizeof(BigArgs)));
  HIPC(hipMemcpy(dev_args, &a, sizeof(BigArgs), hipMemcpyHostToDevice));
  if (variant == 2)
    HIPC(hipFuncSetAttribute((const void*)probe_bigargs_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, PROBE_LDS));
  // warm-up (code object load, LDS opt-in), then eager passes
  for (int i = 0; i < nodes; ++i)
    if ((rc = launch_variant(variant, st, a, dev_args, sink))) goto done;
  HIPC(hipStreamSynchronize(st));
  HIPC(hipEventRecord(e0, st));
  for (int r = 0; r < reps; ++r)
    for (int i = 0; i < nodes; ++i)
      if ((rc = launch_variant(variant, st, a, dev_args, sink))) goto done;
  HIPC(hipEventRecord(e1, st));
  HIPC(hipEventSynchronize(e1));
  HIPC(hipEventElapsedTime(&ms, e0, e1));
  *eager_ms = ms / reps;
  // the same chain captured once and replayed
  HIPC(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < nodes; ++i)
    if ((rc = launch_variant(variant, st, a, dev_args, sink))) {
      hipGraph_t dead = nullptr;
      (void)hipStreamEndCapture(st, &dead);
      if (dead) (void)hipGraphDestroy(dead);
      goto done;
    }
  HIPC(hipStreamEndCapture(st, &graph));
  HIPC(hipGraphGetNodes(graph, nullptr, &nn));
  if (graph_nodes) *graph_nodes = (int32_t)nn;
  HIPC(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  HIPC(hipGraphLaunch(exec, st));                 // first replay: uploads, not timed
  HIPC(hipStreamSynchronize(st));
  HIPC(hipEventRecord(e0, st));
  for (int r = 0; r < reps; ++r) HIPC(hipGraphLaunch(exec, st));
  HIPC(hipEventRecord(e1, st));
  HIPC(hipEventSynchronize(e1));
  HIPC(hipEventElapsedTime(&ms, e0, e1));
  *graph_ms = ms / reps;
done:
  if (exec) (void)hipGraphExecDestroy(exec);
  if (graph) (void)hipGraphDestroy(graph);
  if (sink) (void)hipFree(sink);
  if (dev_args) (void)hipFree(dev_args);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (st) (void)hipStreamDestroy(st);
  return rc;
}

// What the runtime says about the device and itself, as one JSON object (the facts a slow replay is compared against).
extern "C" int gct_device_facts(char* buf, int cap) {
  GCT_CHECK_ARG(buf && cap >= 256, "device_facts: buffer too small");
  int dev = 0, rt = 0, drv = 0;
  (void)hipGetDevice(&dev);
  (void)hipRuntimeGetVersion(&rt);
  (void)hipDriverGetVersion(&drv);
  hipDeviceProp_t p;
  memset(&p, 0, sizeof(p));
  if (hipGetDeviceProperties(&p, dev) != hipSuccess) {
    gct_set_error("device_facts: hipGetDeviceProperties failed");
    return GCT_ERR_HIP;
  }
  int large_bar = -1, host_native_atomic = -1, can_map_host = -1, direct_managed = -1, pageable = -1;
  (void)hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, dev);
  (void)hipDeviceGetAttribute(&host_native_atomic, hipDeviceAttributeHostNativeAtomicSupported, dev);
  (void)hipDeviceGetAttribute(&can_map_host, hipDeviceAttributeCanMapHostMemory, dev);
  (void)hipDeviceGetAttribute(&direct_managed, hipDeviceAttributeDirectManagedMemAccessFromHost, dev);
  (void)hipDeviceGetAttribute(&pageable, hipDeviceAttributePageableMemoryAccess, dev);
  size_t free_b = 0, total_b = 0;
  (void)hipMemGetInfo(&free_b, &total_b);
  snprintf(buf, (size_t)cap,
           "{\"device\": %d, \"name\": \"%s\", \"gcn_arch\": \"%s\", \"hip_runtime_version\": %d, \"hip_driver_version\": %d, "
           "\"compute_units\": %d, \"clock_khz\": %d, \"mem_clock_khz\": %d, \"pci_bus\": \"%04x:%02x:%02x\", "
           "\"is_large_bar\": %d, \"host_native_atomics\": %d, \"can_map_host_memory\": %d, "
           "\"direct_managed_access_from_host\": %d, \"pageable_memory_access\": %d, \"mem_total_gib\": %.1f, "
           "\"mem_free_gib\": %.1f, \"max_shared_per_block\": %zu}",
           dev, p.name, p.gcnArchName, rt, drv, p.multiProcessorCount, p.clockRate, p.memoryClockRate, p.pciDomainID,
           p.pciBusID, p.pciDeviceID, large_bar, host_native_atomic, can_map_host, direct_managed, pageable,
           (double)total_b / (1 << 30), (double)free_b / (1 << 30), p.sharedMemPerBlock);
  return GCT_OK;
}

// Node census of a captured graph (hipGraph_t): kernel / memcpy / memset / other node counts, the largest dynamic-LDS
// request and the largest grid among its kernel nodes.  out8: {nodes, kernels, memcpys, memsets, others, max_lds_bytes,
// max_grid_blocks, kernels_with_lds_over_64k}
extern "C" int gct_graph_census(void* hip_graph, int64_t* out8) {
  GCT_CHECK_ARG(hip_graph && out8, "graph_census: null");
  hipGraph_t g = (hipGraph_t)hip_graph;
  size_t n = 0;
  if (hipGraphGetNodes(g, nullptr, &n) != hipSuccess) {
    gct_set_error("graph_census: hipGraphGetNodes failed");
    return GCT_ERR_HIP;
  }
  for (int i = 0; i < 8; ++i) out8[i] = 0;
  out8[0] = (int64_t)n;
  if (n == 0) return GCT_OK;
  hipGraphNode_t* nodes = (hipGraphNode_t*)malloc(n * sizeof(hipGraphNode_t));
  if (!nodes) {
    gct_set_error("graph_census: out of host memory");
    return GCT_ERR_ARG;
  }
  if (hipGraphGetNodes(g, nodes, &n) != hipSuccess) {
    free(nodes);
    gct_set_error("graph_census: hipGraphGetNodes failed");
    return GCT_ERR_HIP;
  }
  for (size_t i = 0; i < n; ++i) {
    hipGraphNodeType t;
    if (hipGraphNodeGetType(nodes[i], &t) != hipSuccess) { ++out8[4]; continue; }
    if (t == hipGraphNodeTypeKernel) {
      ++out8[1];
      hipKernelNodeParams kp;
      memset(&kp, 0, sizeof(kp));
      if (hipGraphKernelNodeGetParams(nodes[i], &kp) == hipSuccess) {
        if ((int64_t)kp.sharedMemBytes > out8[5]) out8[5] = kp.sharedMemBytes;
        const int64_t blocks = (int64_t)kp.gridDim.x * kp.gridDim.y * kp.gridDim.z;
        if (blocks > out8[6]) out8[6] = blocks;
        if (kp.sharedMemBytes > 64 * 1024) ++out8[7];
      }
    } else if (t == hipGraphNodeTypeMemcpy) {
      ++out8[2];
    } else if (t == hipGraphNodeTypeMemset) {
      ++out8[3];
    } else {
      ++out8[4];
    }
  }
  free(nodes);
  return GCT_OK;
}

// ------------------------------------------------------------------------------------------------ box calibration
// The boxes of a pool do not run a matrix-dense loop at the same speed (MI355X_MICROARCH.md, DVFS give-back item 5: 12 %
// between devices for one binary); this round's sessions saw the SAME training step at 47.6 ... 51.3 ms.  The probe gives
// a bench line something to be read against: the bf16 MFMA rate this device sustains on a register-only loop (two waves
// per SIMD on every CU, v_mfma_f32_16x16x32_bf16 on eight independent accumulators, pseudo-random operands -- zeros would
// run at a higher clock), timed with events on a private stream after a warm-up launch.
namespace {
typedef __bf16 probe_bf16x8 __attribute__((ext_vector_type(8)));
typedef float probe_f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(512) void mfma_probe_kernel(float* sink, int iters) {
  probe_bf16x8 a, b;
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int i = 0; i < 8; ++i) {
    s = s * 1664525u + 1013904223u;
    a[i] = (__bf16)((float)((int)(s >> 20) - 2048) * (1.f / 1024.f));
    s = s * 1664525u + 1013904223u;
    b[i] = (__bf16)((float)((int)(s >> 20) - 2048) * (1.f / 1024.f));
  }
  probe_f32x4 acc[8];
  for (int q = 0; q < 8; ++q) acc[q] = (probe_f32x4){0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[q], 0, 0, 0);
  }
  float t = 0.f;
  for (int q = 0; q < 8; ++q) t += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
  if (t == 123456.789f) sink[0] = t;       // never true: keeps the loop alive
}
}  // namespace

extern "C" int gct_mfma_clock_probe(int iters, float* tflops, float* ms_out) {
  GCT_CHECK_ARG(iters > 0 && tflops, "mfma_clock_probe: bad args");
  int dev = 0;
  hipDeviceProp_t pr;
  if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) {
    gct_set_error("mfma_clock_probe: no device");
    return GCT_ERR_HIP;
  }
  const int grid = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 256;
  hipStream_t st;
  hipEvent_t e0, e1;
  float* sink = nullptr;
  if (hipStreamCreate(&st) != hipSuccess || hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess ||
      hipMalloc(&sink, 64) != hipSuccess) {
    gct_set_error("mfma_clock_probe: cannot create stream / events");
    return GCT_ERR_HIP;
  }
  float best = 1e30f;
  for (int rep = 0; rep < 4; ++rep) {      // rep 0 warms up
    (void)hipEventRecord(e0, st);
    hipLaunchKernelGGL(mfma_probe_kernel, dim3((unsigned)grid), dim3(512), 0, st, sink, iters);
    (void)hipEventRecord(e1, st);
    if (hipEventSynchronize(e1) != hipSuccess) {
      gct_set_error("mfma_clock_probe: launch failed: %s", hipGetErrorString(hipGetLastError()));
      return GCT_ERR_HIP;
    }
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  (void)hipFree(sink);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipStreamDestroy(st);
  const double flop = (double)grid * 8.0 * (double)iters * 8.0 * 16384.0;     // blocks x waves x iterations x MFMAs x flop
  *tflops = (float)(flop / ((double)best * 1e-3) / 1e12);
  if (ms_out) *ms_out = best;
  return GCT_OK;
}
