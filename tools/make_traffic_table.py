#!/usr/bin/env python3
"""HBM-side traffic per launch from the rocprofv3 PMC passes of tools/profile_gpu.sh -> profiles/r06_traffic.json.

FETCH_SIZE and WRITE_SIZE are collected in SEPARATE passes (TCC slots, MI355X_MICROARCH.md "rocprofv3 PMC slots"); both
are reported in KiB.  gfx950 correction from the same guide (HBM section): FETCH_SIZE counts 64 B per 128-B request for
wide coalesced streams (16 B per lane, LDS-DMA alike) -> doubled; WRITE_SIZE is exact for 16-B-per-lane stores.
The table records the sha256 of the libyolov3_hip.so that was profiled; bench.py only uses a table whose hash matches
the library it has loaded (otherwise roofline.traffic is null).

Usage (on the GPU box, after tools/profile_gpu.sh <tag> ...):  python3 tools/make_traffic_table.py gpurun_out/prof_<tag> <out.json>
"""
import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


DT = {"DF16b": "bf16", "__bf16": "bf16", "DF16_": "f16", "_Float16": "f16", "f": "f32", "float": "f32"}
_T = r"(DF16b|DF16_|f)"                      # Itanium-mangled element type: __bf16, _Float16, float


def plan_name(kernel):
    """rocprofv3's kernel name -> the name the plan executor reports (y3_plan_op_kernel).  rocprofv3 leaves the template
    kernels of anonymous namespaces mangled (...conv_halo_ws_kernelIDF16bLi4ELi4EEEv...); since round 5 every 16-bit kernel
    is a template on its element type (DF16b = bf16, DF16_ = fp16)."""
    k = kernel.replace("(anonymous namespace)::", "").replace("void ", "")
    m = re.match(r"conv_halo_ws_kernel<(__bf16|_Float16|float), \d+, (\d+)>", k)
    if m:
        return "conv_halo_ws_%s_%dx128" % (DT[m.group(1)], 64 * int(m.group(2)))
    m = re.search(r"conv_halo_ws_kernelI" + _T + r"Li\d+ELi(\d+)E", k)
    if m:
        return "conv_halo_ws_%s_%dx128" % (DT[m.group(1)], 64 * int(m.group(2)))
    m = re.search(r"conv_halo_dw_kernelI" + _T, k) or re.match(r"conv_halo_dw_kernel<(__bf16|_Float16)", k)
    if m:
        return "conv_halo_dw_%s_192x256" % DT[m.group(1)]
    m = re.search(r"conv_igemm2_kernelI" + _T + r"Li64ELi256ELi1ELi4ELi0ELb1E", k)
    if m:
        return "conv_head_decode_%s_64x256" % DT[m.group(1)]
    m = re.search(r"conv_igemm(2|3|)_kernelI" + _T + r"Li(\d+)ELi(\d+)E", k)
    if m:
        return "conv_igemm%s_%s_%sx%s" % (m.group(1), DT[m.group(2)], m.group(3), m.group(4))
    m = re.match(r"conv_igemm(2|3|)_kernel<(float|__bf16|_Float16), (\d+), (\d+)", k)     # demangled form
    if m:
        return "conv_igemm%s_%s_%sx%s" % (m.group(1), DT[m.group(2)], m.group(3), m.group(4))
    m = re.match(r"conv_patch_wsp_kernel<(float|__bf16|_Float16)", k)
    if m:
        return "conv_patch_wsp_%s_8x32x128" % DT[m.group(1)]
    m = re.search(r"conv_patch_wsp_kernelI" + _T, k)
    if m:
        return "conv_patch_wsp_%s_8x32x128" % DT[m.group(1)]
    m = re.search(r"conv1x1_dw_kernelI" + _T + r"Li(\d+)ELi\d+ELb1E", k) or re.match(r"conv1x1_dw_kernel<(__bf16|_Float16), (\d+), \d+, true", k)
    if m:                                      # HEAD = true: the detection-head form
        return "conv_head_decode_dw_%s_%sx256" % (DT[m.group(1)], m.group(2))
    m = re.search(r"conv1x1_dw_kernelI" + _T + r"Li(\d+)E", k) or re.match(r"conv1x1_dw_kernel<(__bf16|_Float16), (\d+)", k)
    if m:
        return "conv1x1_dw_%s_%sx256" % (DT[m.group(1)], m.group(2))
    m = re.search(r"conv1x1_wres_kernelI" + _T + r"Li(\d+)E", k)
    if m:
        return "conv1x1_wres_%s_128x%s" % (DT[m.group(1)], m.group(2))
    m = re.search(r"conv_stem_s2_(ws|fused)_kernelI" + _T, k)
    if m:
        return "conv_stem_s2_fused_u8_%s" % DT[m.group(2)]
    m = re.search(r"conv_resblock_fused_kernelI" + _T, k)
    if m:
        return "conv_resblock_fused_%s_64_32_64" % DT[m.group(1)]
    m = re.search(r"conv_block_fused_kernelI" + _T, k)
    if m:
        return "conv_block_fused_%s_x128" % DT[m.group(1)]
    m = re.search(r"\d+(conv_[a-z0-9_]+_kernel|[a-z_]+_kernel)I", k)
    if m:
        return m.group(1) + k[k.index(m.group(1)) + len(m.group(1)):].split("EvN")[0]
    return k.split("(")[0]


def per_launch(path, counter):
    acc, n = defaultdict(float), defaultdict(int)
    with open(path) as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] != counter:
                continue
            k = plan_name(r["Kernel_Name"])
            acc[k] += float(r["Counter_Value"])
            n[k] += 1
    return {k: (acc[k] / n[k], n[k]) for k in acc}


WORKLOADS = ["yolov3_608_b16_bf16", "yolov3-tiny_416_b8_float32", "yolov3-spp_608_b16_bf16", "yolov3_608_b16_float32",
             "yolov3_608_b16_fp16"]


def kernels_of(prof, suffix):
    fetch = glob.glob(os.path.join(prof, "pmc_fetch" + suffix, "**", "*counter_collection.csv"), recursive=True)
    write = glob.glob(os.path.join(prof, "pmc_write" + suffix, "**", "*counter_collection.csv"), recursive=True)
    if not fetch or not write:
        return None
    f, w = per_launch(fetch[0], "FETCH_SIZE"), per_launch(write[0], "WRITE_SIZE")
    out = {}
    for k in sorted(set(f) & set(w)):
        out[k] = {"fetch_size_kib": round(f[k][0], 1), "write_size_kib": round(w[k][0], 1), "launches": f[k][1],
                  "traffic_bytes_per_launch": int(round((2.0 * f[k][0] + w[k][0]) * 1024.0))}
    return out


def trace_durations(prof):
    """{plan kernel name: (mean us, median us, launches)} from the rocprofv3 kernel trace of the headline workload
    (tools/profile_gpu.sh: <prof>/trace): the figure bench.py's dispatch-bound events must agree with (roofline.rocprof_avg_us)."""
    hits = glob.glob(os.path.join(prof, "trace", "**", "*kernel_trace.csv"), recursive=True)
    if not hits:
        return {}
    dur = defaultdict(list)
    with open(hits[0]) as fh:
        for r in csv.DictReader(fh):
            dur[plan_name(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    out = {}
    for k, v in dur.items():
        v = sorted(v)
        out[k] = (sum(v) / len(v), v[len(v) // 2], len(v))
    return out


def main():
    prof, out = sys.argv[1], sys.argv[2]
    with open(os.path.join(ROOT, "pytorch-yolov3_amd", "lib", "libyolov3_hip.so"), "rb") as fh:
        sha = hashlib.sha256(fh.read()).hexdigest()
    sys.path.insert(0, ROOT)
    import bench
    table = {"lib_sha256": sha,
             "device_code_sha256": bench.device_code_sha256(os.path.join(ROOT, "pytorch-yolov3_amd", "lib", "libyolov3_hip.so")),
             "_source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/profile_gpu.sh) over "
                        "bench.py --streams 1 with the benchmark's plan options, once per workload (a kernel's mean traffic "
                        "depends on the layers it runs); per-launch means; FETCH_SIZE x 2 (gfx950 counts 64 B per 128-B request "
                        "for 16-B-per-lane streams: MI355X_MICROARCH.md, HBM), WRITE_SIZE as read; KiB -> bytes",
             "workloads": {}}
    main_k = kernels_of(prof, "")
    if main_k is None:
        raise SystemExit("no PMC csv under %s" % prof)
    for k, (mean_us, med_us, n) in trace_durations(prof).items():
        if k in main_k:
            main_k[k].update(rocprof_avg_us=round(mean_us, 2), rocprof_median_us=round(med_us, 2), rocprof_launches=n)
    table["workloads"][WORKLOADS[0]] = {"kernels": main_k}
    table["kernels"] = main_k                            # the headline workload, also at the top level
    for wk in WORKLOADS[1:]:
        k = kernels_of(prof, "_" + wk)
        if k is not None:
            table["workloads"][wk] = {"kernels": k}
    with open(out, "w") as fh:
        json.dump(table, fh, indent=1, sort_keys=True)
    for wk, t in table["workloads"].items():
        print("==", wk)
        for k, v in sorted(t["kernels"].items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"])[:8]:
            print("%-48s launches %5d  traffic/launch %8.1f MB" % (k[:48], v["launches"], v["traffic_bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main()
