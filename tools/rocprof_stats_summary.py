"""Reduce a rocprofv3 `--kernel-trace --stats --output-format csv` kernel_stats.csv of `bench.py --timed-only` to the per-family
table the bench line is checked against: every profiled step is the timed workload (warmup + steps + the one HIP-event-profiled
step), so total / steps / launches-per-step is the average launch duration of a family in the timed region.

usage: python tools/rocprof_stats_summary.py <kernel_stats.csv> <steps_in_run> > profiles/rNN_rocprofv3_family_summary.txt"""
import csv, sys, collections

FAMILIES = [("conv3x3 (conv_mfma_kernel<9,...>)", "conv_mfma_kernel<9"), ("gemm / 1x1 (conv_mfma_kernel<1,...>)", "conv_mfma_kernel<1"), ("gemm, plane-fed (gemm_p3_kernel)", "gemm_p3_kernel"),
            ("attn_d64 (incl. the pipeline kernels)", "attn_d64_"), ("attn_d512", "attn_d512_kernel"), ("transpose_v", "transpose_v_kernel"),
            ("gn_stats / partials", "gn_"), ("layernorm", "layernorm_")]
steps = int(sys.argv[2])
agg = collections.OrderedDict((f[0], [0, 0.0]) for f in FAMILIES)
other = [0, 0.0]
for r in csv.DictReader(open(sys.argv[1])):
    name, calls, tot = r["Name"], int(r["Calls"]), float(r["TotalDurationNs"])
    for label, key in FAMILIES:
        if key in name:
            agg[label][0] += calls; agg[label][1] += tot
            break
    else:
        other[0] += calls; other[1] += tot
print(f"{'family':44s} {'launches/step':>13s} {'ms/step':>10s} {'avg launch us':>14s}")
tot_ms = 0.0
for label, (calls, ns) in list(agg.items()) + [("other", other)]:
    if calls == 0:
        continue
    tot_ms += ns / 1e6 / steps
    print(f"{label:44s} {calls / steps:13.1f} {ns / 1e6 / steps:10.3f} {ns / 1e3 / calls:14.2f}")
print(f"{'sum of kernel time per step':44s} {'':13s} {tot_ms:10.3f}")
