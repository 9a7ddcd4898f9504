#!/usr/bin/env python3
"""Regenerate the measured artifacts under profiles/ for the current sources, on the MI355X box, in one go:

    python3 tools/refresh_profiles.py [--round r04] [--skip name,...] [--only name,...]

  bench        python bench.py --steps 20 --warmup 5                       -> <round>_bench_n1.json
  trace        rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 5 (same arguments as the driver's run)
                                                                           -> <round>_bench_default_kernel_stats.csv, <round>_bench_default_traced.json
  pmc          tools/collect_traffic.py (per-XCD decoders, 32 utterances)  -> <round>_pmc_traffic.json
  pmc_big      tools/collect_traffic.py --utterances 128 --mode graph16    -> <round>_pmc_traffic_big128.json
  pmc_xcm      tools/collect_traffic.py --utterances 128 --mode xcm        -> <round>_pmc_traffic_xcm128.json
  pmc_sq       tools/collect_sq.py --mode xcd (32 utterances), --mode xcm (128): SQ counters (wave / busy / wait cycles, VALU / MFMA /
               LDS instructions, MFMA-busy cycles)                         -> <round>_pmc_sq_xcd32.json, <round>_pmc_sq_xcm128.json
  manifest     python bench.py --workload manifest --force-gather (BASELINE configs[4] through shard.convert_sharded, the ragged
               gather on RCCL at world size 1)                             -> <round>_bench_manifest_rccl_world1.json
  encoder      rocprofv3 --kernel-trace --stats -- python3 tools/profile_encoder.py [c1]
                                                                           -> <round>_encoder_c2_kernel_stats.csv, <round>_encoder_c1_kernel_stats.csv
  timeline     tools/xcd_timeline.py 1 8 32, tools/xcm_timeline.py 128 (debug build with stamps)
                                                                           -> <round>_xcd_timeline.txt, <round>_xcm_timeline.txt
  probe        tools/xcd_decoder_probe.py, tools/xcm_probe.py              -> <round>_xcd_probe.csv, <round>_xcm_probe.csv
  by_batch     tools/bench_by_batch.py                                     -> <round>_bench_by_batch.csv
  summary      rewrites the numbers quoted in profiles/<round>_summary.md from the files above

Every step is a fresh child process; the parent never touches the GPU (rocprofv3 must have the program itself after `--`).
The artifacts are written to gpurun_out/profiles_<round>/ (the only directory that travels back from the GPU box): copy them
into profiles/ and commit.  Scratch (kernel traces) goes to /tmp.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PY = sys.executable


def run(cmd, out=None, cwd=ROOT, timeout=420, env=None):
    print("[refresh]", " ".join(cmd), flush=True)
    e = dict(os.environ, TMPDIR="/tmp")
    e.update(env or {})
    with open(out, "w") if out else open(os.devnull, "w") as f:
        rc = subprocess.run(cmd, cwd=cwd, stdout=f if out else None, stderr=subprocess.PIPE if out else None, timeout=timeout, env=e)
    if rc.returncode != 0:
        print("[refresh] FAILED (%d): %s" % (rc.returncode, (rc.stderr or b"").decode()[-2000:]), flush=True)
    return rc.returncode == 0


def kernel_stats(scratch):
    for path in glob.glob(os.path.join(scratch, "**", "*kernel_stats.csv"), recursive=True):
        return path
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r04")
    ap.add_argument("--skip", default="")
    ap.add_argument("--only", default="")
    args = ap.parse_args()
    skip = set(s for s in args.skip.split(",") if s)
    if args.only:
        every = {"bench", "trace", "pmc", "pmc_big", "pmc_xcm", "pmc_sq", "manifest", "encoder", "timeline", "probe", "by_batch", "summary"}
        skip = every - set(s for s in args.only.split(",") if s)
    R = args.round
    # on the GPU box only gpurun_out/ travels back: the artifacts go to gpurun_out/profiles_<round>/ (copy them into profiles/
    # afterwards), the bulky scratch (kernel traces) to /tmp
    P = os.path.join(ROOT, "gpurun_out", f"profiles_{R}")
    S = os.path.join("/tmp", "vqcpc_refresh")
    os.makedirs(P, exist_ok=True)
    os.makedirs(S, exist_ok=True)
    # steps that are skipped keep their committed artifact (the summary is written from the files of P)
    for path in glob.glob(os.path.join(ROOT, "profiles", f"{R}_*")):
        if not os.path.exists(os.path.join(P, os.path.basename(path))):
            shutil.copy(path, P)
    bench_args = ["--steps", "20", "--warmup", "5"]
    ok = {}

    if "bench" not in skip:
        ok["bench"] = run([PY, "bench.py"] + bench_args, out=os.path.join(P, f"{R}_bench_n1.json"))
    if "trace" not in skip:
        d = os.path.join(S, "trace")
        shutil.rmtree(d, ignore_errors=True)
        good = run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--", PY, os.path.join(ROOT, "bench.py")] +
                   bench_args + ["--no-cpu-baseline"], out=os.path.join(P, f"{R}_bench_default_traced.json"), cwd="/tmp")
        ks = kernel_stats(d)
        if good and ks:
            shutil.copy(ks, os.path.join(P, f"{R}_bench_default_kernel_stats.csv"))
        ok["trace"] = bool(good and ks)
    if "pmc" not in skip:
        ok["pmc"] = run([PY, "tools/collect_traffic.py", "--mode", "xcd", "--utterances", "32", "--out", os.path.join(P, f"{R}_pmc_traffic.json")],
                        out=os.path.join(S, "pmc.log"))
    if "pmc_big" not in skip:
        ok["pmc_big"] = run([PY, "tools/collect_traffic.py", "--mode", "graph16", "--utterances", "128",
                             "--out", os.path.join(P, f"{R}_pmc_traffic_big128.json")], out=os.path.join(S, "pmc_big.log"))
    if "pmc_xcm" not in skip:
        ok["pmc_xcm"] = run([PY, "tools/collect_traffic.py", "--mode", "xcm", "--utterances", "128",
                             "--out", os.path.join(P, f"{R}_pmc_traffic_xcm128.json")], out=os.path.join(S, "pmc_xcm.log"))
    if "pmc_sq" not in skip:
        ok["pmc_sq_xcd"] = run([PY, "tools/collect_sq.py", "--mode", "xcd", "--utterances", "32", "--out", os.path.join(P, f"{R}_pmc_sq_xcd32.json")],
                               out=os.path.join(S, "pmc_sq_xcd.log"), timeout=900)
        ok["pmc_sq_xcm"] = run([PY, "tools/collect_sq.py", "--mode", "xcm", "--utterances", "128", "--out", os.path.join(P, f"{R}_pmc_sq_xcm128.json")],
                               out=os.path.join(S, "pmc_sq_xcm.log"), timeout=900)
    if "manifest" not in skip:
        ok["manifest"] = run([PY, "bench.py", "--workload", "manifest", "--manifest", "512", "--force-gather", "--steps", "3", "--warmup", "1"],
                             out=os.path.join(P, f"{R}_bench_manifest_rccl_world1.json"))
    if "encoder" not in skip:
        for case, extra in (("c2", []), ("c1", ["c1"])):
            d = os.path.join(S, "enc_" + case)
            shutil.rmtree(d, ignore_errors=True)
            good = run(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", d, "--", PY,
                        os.path.join(ROOT, "tools", "profile_encoder.py")] + extra, out=os.path.join(S, f"enc_{case}.log"), cwd="/tmp")
            ks = kernel_stats(d)
            if good and ks:
                shutil.copy(ks, os.path.join(P, f"{R}_encoder_{case}_kernel_stats.csv"))
            ok["encoder_" + case] = bool(good and ks)
    if "timeline" not in skip:
        ok["timeline"] = (run(["sh", "tools/build_stamps.sh"]) and
                          run([PY, "tools/xcd_timeline.py", "1", "8", "32"], out=os.path.join(P, f"{R}_xcd_timeline.txt")) and
                          run([PY, "tools/xcm_timeline.py", "128"], out=os.path.join(P, f"{R}_xcm_timeline.txt")))
    if "probe" not in skip:
        ok["probe"] = run([PY, "tools/xcd_decoder_probe.py"], out=os.path.join(S, "probe.log"))
        src = os.path.join(ROOT, "gpurun_out", "xcd_probe.csv")
        if ok["probe"] and os.path.exists(src):
            shutil.copy(src, os.path.join(P, f"{R}_xcd_probe.csv"))
        ok["probe_xcm"] = run([PY, "tools/xcm_probe.py"], out=os.path.join(P, f"{R}_xcm_probe.csv"))
    if "by_batch" not in skip:
        ok["by_batch"] = run([PY, "tools/bench_by_batch.py", "1", "8", "16", "32", "64", "96", "128", "256", "384", "512"],
                             out=os.path.join(P, f"{R}_bench_by_batch.csv"))
    if "summary" not in skip:
        summary(P, R)
    print("[refresh]", json.dumps(ok), flush=True)
    return 0 if all(ok.values()) else 1


def summary(P, R):
    """The numbers DESIGN.md quotes, copied from the committed files (so that the text and the files cannot drift apart)."""
    lines = [f"# {R}: numbers quoted in DESIGN.md, generated by tools/refresh_profiles.py from the files of this directory", ""]
    try:
        b = json.load(open(os.path.join(P, f"{R}_bench_n1.json")))
        r = b["roofline"]
        lines += [f"* `{R}_bench_n1.json` (`python bench.py --steps 20 --warmup 5`): value {b['value']:.0f} samples/s, {b['ms_per_step']:.2f} ms per step; "
                  f"roofline: {r['kernel'].split(' (')[0]}, launch {r['avg_launch_us']:.1f} us, achieved {r['achieved']:.2f} TFLOP/s = frac {r['frac']:.4f} "
                  f"(executed {r.get('frac_executed', float('nan')):.4f}); decode step {r['decode_step']['us']:.3f} us; traffic {r.get('traffic')} ({r.get('traffic_source')})"]
        for k in ("single_utterance", "one_gpu_256", "manifest", "convert_e2e", "teacher_forced", "encoder"):
            if k in b:
                v = b[k]
                keep = {a: (round(c, 4) if isinstance(c, float) else c) for a, c in v.items() if not isinstance(c, (dict, list, str))}
                lines.append(f"* `{k}`: {keep}")
        if "one_gpu_256" in b:
            rr = b["one_gpu_256"]["roofline"]
            lines.append(f"* `one_gpu_256.roofline`: {rr['kernel'].split(':')[0].split(' (')[0]}, launch {rr['avg_launch_us']:.1f} us, frac {rr['frac']:.4f}")
        if "cpu_baseline" in b:
            c = b["cpu_baseline"]
            lines.append(f"* `cpu_baseline`: {c['value']:.0f} samples/s on {c['cores']} threads ({c['kind']}); encoder {c['encoder_frames_per_s']:.0f} frames/s")
    except (OSError, ValueError, KeyError) as e:
        lines.append(f"* bench file missing or incomplete: {e}")
    try:
        with open(os.path.join(P, f"{R}_bench_default_kernel_stats.csv"), newline="") as f:
            rows = list(csv.DictReader(f))
        lines.append(f"* `{R}_bench_default_kernel_stats.csv` (rocprofv3 --kernel-trace --stats of the same command): top kernels by total time:")
        for row in rows[:6]:
            lines.append(f"    * {row['Name'][:90]}: calls {row['Calls']}, average {float(row['AverageNs']) / 1e3:.2f} us, {row['Percentage']} %")
    except (OSError, ValueError, KeyError) as e:
        lines.append(f"* kernel stats missing: {e}")
    for name in (f"{R}_pmc_traffic.json", f"{R}_pmc_traffic_big128.json", f"{R}_pmc_traffic_xcm128.json"):
        try:
            t = json.load(open(os.path.join(P, name)))
            lines.append(f"* `{name}`: {t['kernel'][:70]}, {t['utterances']} utterances: traffic {t['traffic_bytes_per_launch'] / 1e6:.2f} MB per launch, "
                         f"algorithmic {t['algorithmic_bytes_per_launch'] / 1e6:.2f} MB, ratio {t['traffic_over_algorithmic']:.2f}, L2 hit rate {t['l2_hit_rate']:.3f}")
        except (OSError, ValueError, KeyError) as e:
            lines.append(f"* `{name}` missing: {e}")
    for name in (f"{R}_pmc_sq_xcd32.json", f"{R}_pmc_sq_xcm128.json"):
        try:
            q = json.load(open(os.path.join(P, name)))
            d = q["derived"]
            lines.append(f"* `{name}`: {q['kernel'][:60]}, {q['utterances']} utterances x {q['sample_steps']} steps: "
                         f"MFMA-busy / kernel clocks {d.get('mfma_busy_over_kernel_clocks', float('nan')):.3f} "
                         f"({d.get('mfma_busy_clocks_per_simd_per_step', float('nan')):.0f} clocks per SIMD and step of {d.get('clocks_per_sample_step', float('nan')):.0f}), "
                         f"per wave and step: {d.get('SQ_INSTS_VALU_per_wave_per_step', float('nan')):.0f} VALU, {d.get('SQ_INSTS_MFMA_per_wave_per_step', float('nan')):.0f} MFMA, "
                         f"{d.get('SQ_INSTS_LDS_per_wave_per_step', float('nan')):.0f} LDS instructions; of the wave cycles: active {d.get('SQ_ACTIVE_INST_ANY_over_WAVE_CYCLES', float('nan')):.3f}, "
                         f"waiting (s_waitcnt / barrier) {d.get('SQ_WAIT_ANY_over_WAVE_CYCLES', float('nan')):.3f}, issue-stalled {d.get('SQ_WAIT_INST_ANY_over_WAVE_CYCLES', float('nan')):.3f}; "
                         f"{q.get('us_per_step_profiled')} us per step under the profiler")
        except (OSError, ValueError, KeyError) as e:
            lines.append(f"* `{name}` missing: {e}")
    try:
        m = json.load(open(os.path.join(P, f"{R}_bench_manifest_rccl_world1.json")))
        lines.append(f"* `{R}_bench_manifest_rccl_world1.json` (`python bench.py --workload manifest --force-gather`): {m['value']:.0f} samples/s, "
                     f"{m['config']['utterances']} utterances, backend {m['backend']} x {m['rccl_ranks']}, ragged gather {m['gather']['ms_rank0_last_step']:.2f} ms, "
                     f"all gathered: {m['all_utterances_gathered_with_their_lengths']}, work space {m.get('workspace_peak_bytes_rank0', 0) / 1e9:.2f} GB")
    except (OSError, ValueError, KeyError) as e:
        lines.append(f"* manifest (configs[4]) line missing: {e}")
    with open(os.path.join(P, f"{R}_summary.md"), "w") as f:
        f.write("\n".join(lines) + "\n")


if __name__ == "__main__":
    sys.exit(main())
