"""What this box does with hipGraph replay (libgctplus_diag.so = csrc/graphprobe.hip, include/gctplus_diag.h -- a
diagnostics library beside the operator library, not part of its ABI): facts, micro-probes and the census of a captured
graph.  Used by KVDecoder's replay guard, bench.py's decode block and tools/graph_probe.py.

Why it exists: BASELINE configs[4] asks for a hipGraph-captured decode step (reference loop:
Inference/sampling_tool.py:140-184); on some boxes of this pool replaying the captured step was 3-13x slower than
launching the same ~70 kernels one by one (BENCH_r02.json: 31.4 vs 2.46 ms per token).  Nothing here changes what
is computed -- it only tells the caller which launch mode is the fast one and records the evidence.
"""
from __future__ import annotations

import ctypes
import glob
import json
import os

from ._lib import check_diag as check, load_diag as load

VARIANTS = {0: "tiny kernel, 8-byte kernarg, 1 workgroup",
            1: "320-byte by-value kernarg read by every wave, 2048 x 512 threads",
            2: "as 1 + 144 KB dynamic LDS",
            3: "as 1, arguments behind one pointer into device memory"}

# runtime switches that change how graph nodes and their kernel arguments are issued (strings of libamdhip64.so, ROCm 7)
RUNTIME_KNOBS = ["HIP_FORCE_DEV_KERNARG", "DEBUG_CLR_GRAPH_PACKET_CAPTURE", "DEBUG_HIP_GRAPH_BATCH_SIZE",
                 "DEBUG_HIP_FORCE_GRAPH_QUEUES", "DEBUG_HIP_KERNARG_COPY_OPT", "DEBUG_CLR_KERNARG_HDP_FLUSH_WA",
                 "ROC_ENABLE_LARGE_BAR", "ROC_USE_FGS_KERNARG", "AMD_DIRECT_DISPATCH", "GPU_MAX_HW_QUEUES",
                 "HIP_LAUNCH_BLOCKING", "AMD_SERIALIZE_KERNEL", "HSA_ENABLE_SDMA", "HSA_ENABLE_INTERRUPT"]


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def device_facts() -> dict:
    """Runtime / driver versions, large-BAR and host-access attributes, PCIe link, kernel driver version, the runtime
    switches set in this process's environment."""
    buf = ctypes.create_string_buffer(2048)
    check(load().gct_device_facts(buf, len(buf)), "gct_device_facts")
    facts = json.loads(buf.value.decode())
    bus = facts.get("pci_bus")
    sysdev = f"/sys/bus/pci/devices/{bus}.0" if bus else None
    if sysdev and os.path.isdir(sysdev):
        for key in ("current_link_speed", "current_link_width", "max_link_speed", "max_link_width"):
            facts["pcie_" + key] = _read(os.path.join(sysdev, key))
        res = _read(os.path.join(sysdev, "resource"))
        if res:                                             # BAR sizes: a large-BAR card exposes all of its VRAM
            sizes = []
            for line in res.splitlines()[:6]:
                a, b, _ = (int(x, 16) for x in line.split())
                sizes.append(0 if b == 0 else (b - a + 1) >> 20)
            facts["pci_bar_mib"] = sizes
    facts["amdgpu_module_version"] = _read("/sys/module/amdgpu/version")
    facts["kernel"] = _read("/proc/sys/kernel/osrelease")
    facts["iommu_groups"] = len(glob.glob("/sys/kernel/iommu_groups/*"))
    facts["env"] = {k: os.environ[k] for k in RUNTIME_KNOBS if k in os.environ}
    return facts


def micro_probe(nodes: int = 70, reps: int = 20, variants=(0, 1, 2, 3)) -> dict:
    """ms per pass of `nodes` chained do-nothing launches, eager vs replayed, per variant (VARIANTS)."""
    out = {}
    for v in variants:
        e, g, n = ctypes.c_float(), ctypes.c_float(), ctypes.c_int32()
        check(load().gct_graph_probe(v, nodes, reps, ctypes.byref(e), ctypes.byref(g), ctypes.byref(n)),
              "gct_graph_probe")
        out[f"v{v}"] = {"what": VARIANTS[v], "eager_ms": round(e.value, 4), "graph_ms": round(g.value, 4),
                        "graph_nodes": n.value, "us_per_node_eager": round(e.value / nodes * 1e3, 2),
                        "us_per_node_graph": round(g.value / nodes * 1e3, 2)}
    return out


def mfma_probe(iters: int = 20000) -> dict:
    """Box calibration (gct_mfma_clock_probe): the bf16 MFMA rate this device sustains on a register-only loop."""
    tf, ms = ctypes.c_float(), ctypes.c_float()
    check(load().gct_mfma_clock_probe(int(iters), ctypes.byref(tf), ctypes.byref(ms)), "gct_mfma_clock_probe")
    return {"bf16_mfma_tflops": round(tf.value, 1), "ms": round(ms.value, 3), "iters": int(iters),
            "what": "register-only v_mfma_f32_16x16x32_bf16 loop, two waves per SIMD on every CU, pseudo-random operands "
                    "(dense peak ~2500): devices of one pool differ by ~10 % here and so does the training step"}


def census(torch_graph) -> dict | None:
    """Node census of a torch.cuda.CUDAGraph created with keep_graph=True (None if the handle is not available)."""
    try:
        h = torch_graph.raw_cuda_graph()
    except Exception:                                        # noqa: BLE001 -- older torch / graph not kept
        return None
    out = (ctypes.c_int64 * 8)()
    check(load().gct_graph_census(ctypes.c_void_p(int(h)), ctypes.cast(out, ctypes.c_void_p)), "gct_graph_census")
    keys = ["nodes", "kernel_nodes", "memcpy_nodes", "memset_nodes", "other_nodes", "max_dynamic_lds_bytes",
            "max_grid_blocks", "kernels_over_64k_lds"]
    return dict(zip(keys, (int(v) for v in out)))
