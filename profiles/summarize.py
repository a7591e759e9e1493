#!/usr/bin/env python3
"""Condense rocprofv3 CSV output into small, committable summaries (profiles/<tag>_*.{csv,json}).

usage: summarize.py <gpurun_out/prof_TAG dir> <TAG>
FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB-like units of the derived metric; on gfx950
FETCH_SIZE under-reports wide coalesced reads by 2x (MI355X_MICROARCH.md "HBM"): the summary stores the
raw counter and the corrected byte count (fetch x 2).
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(d, pat):
    r = sorted(glob.glob(os.path.join(d, "**", pat), recursive=True))
    return r[0] if r else None


def main():
    out_dir, tag = sys.argv[1], sys.argv[2]
    root = os.path.dirname(os.path.abspath(__file__))
    summary = {}
    prev = os.path.join(root, f"{tag}_summary.json")
    if os.path.exists(prev):   # the passes may come from several gpurun calls (the box does not persist): merge into what is there
        with open(prev) as f:
            summary = json.load(f)
    stats = find(os.path.join(out_dir, "trace"), "*kernel_stats.csv")
    if stats:
        rows = list(csv.DictReader(open(stats)))
        keep = rows[:48] + [r for r in rows[48:] if "ccz" in r.get("Name", "")]   # the top of the list + every kernel of this library
        with open(os.path.join(root, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
            w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
            w.writeheader()
            w.writerows(keep)
        for k in ("k_step", "k_softmax_gather", "k_select", "k_expand_backup", "k_finish_move", "k_harvest", "k_bias_act", "k_conv3x3"):
            # every instantiation / variant of a kernel family added up (k_conv3x3: g16 with and without residual, the edge-pair kernel)
            hit = [r for r in rows if k in r.get("Name", "")]
            if hit:
                calls = sum(int(float(r.get("Calls", 0) or 0)) for r in hit)
                tot = sum(float(r.get("TotalDurationNs", 0) or 0) for r in hit)
                summary.setdefault(k, {})["avg_ns"] = tot / max(1, calls)
                summary[k]["calls"] = calls
                summary[k]["pct"] = sum(float(r.get("Percentage", 0) or 0) for r in hit)
                if len(hit) > 1:
                    summary[k]["variants"] = {r["Name"][:80]: {"calls": int(float(r["Calls"])), "avg_ns": float(r["AverageNs"])} for r in hit}
        # the evaluator's NON-tower kernels per step: every planned step launches k_cache_plan exactly once, so a kernel's launches
        # per step = its calls / k_cache_plan's calls (template instantiations of one kernel are added up)
        plan_calls = sum(int(float(r.get("Calls", 0) or 0)) for r in rows if "k_cache_plan" in r.get("Name", ""))
        if plan_calls:
            tail = {}
            for k in ("k_cache_probe", "k_cache_plan", "k_pack_live_planes", "k_head_conv1x1", "k_fc_f16", "k_fc_wide_f16", "k_value_out", "k_softmax_gather"):
                tot = sum(float(r.get("TotalDurationNs", 0) or 0) for r in rows if k in r.get("Name", ""))
                calls = sum(int(float(r.get("Calls", 0) or 0)) for r in rows if k in r.get("Name", ""))
                if calls:
                    tail[k] = {"us_per_step": tot / plan_calls * 1e-3, "launches_per_step": calls / plan_calls, "avg_us": tot / calls * 1e-3}
            # the head convolutions ride in the LAST tower layer's epilogue (k_conv3x3_g16_heads / _edge_heads): what they cost is the
            # difference between that launch and the same launch of an ordinary residual layer (k_conv3x3_g16<true> / _edge<true>)
            def fam(pred):
                hit = [r for r in rows if pred(r.get("Name", ""))]
                calls = sum(int(float(r.get("Calls", 0) or 0)) for r in hit)
                tot = sum(float(r.get("TotalDurationNs", 0) or 0) for r in hit)
                return calls, tot
            extra, detail = 0.0, {}
            for name, heads, plain in (("middle", lambda n: "k_conv3x3_g16_heads" in n, lambda n: "k_conv3x3_g16<true>" in n or "k_conv3x3_g16ILb1" in n),
                                       ("edge", lambda n: "k_conv3x3_g16_edge_heads" in n, lambda n: "k_conv3x3_g16_edge<true>" in n or "k_conv3x3_g16_edgeILb1" in n)):
                hc, ht = fam(heads)
                pc, pt = fam(plain)
                if hc and pc:
                    d = (ht / hc - pt / pc) * hc / plan_calls * 1e-3
                    extra += d
                    detail[name] = {"avg_us_with_heads": ht / hc * 1e-3, "avg_us_plain_residual_layer": pt / pc * 1e-3, "launches_per_step": hc / plan_calls, "extra_us_per_step": d}
            if detail:
                tail["heads_in_last_tower_layer"] = {"us_per_step": extra, "launches_per_step": sum(v["launches_per_step"] for v in detail.values()),
                                                     "avg_us": extra / max(1e-9, sum(v["launches_per_step"] for v in detail.values())), "detail": detail,
                                                     "what": "launch time of the last layer with the heads in its epilogue minus an ordinary residual layer's, x launches per step "
                                                             "(the chains run concurrently: an upper bound on what the step pays)"}
            # whatever torch still launches inside a step (elementwise / GEMM kernels of the library): everything that is not ours
            other = [(r.get("Name", ""), float(r.get("TotalDurationNs", 0) or 0), int(float(r.get("Calls", 0) or 0))) for r in rows
                     if "ccz" not in r.get("Name", "")]   # (ccz:: in demangled names, 3ccz in mangled ones)
            summary["evaluator_tail"] = {"kernels": tail, "us_per_step_ours": sum(v["us_per_step"] for v in tail.values()),
                                         "library_kernels_total_us_per_step": sum(t for _, t, _ in other) / plan_calls * 1e-3,
                                         "library_kernels_top": [{"name": n[:90], "us_per_step": t / plan_calls * 1e-3, "calls": c} for n, t, c in sorted(other, key=lambda x: -x[1])[:8]],
                                         "steps": plan_calls,
                                         "what": "non-tower evaluator work per lockstep step (rocprofv3 --kernel-trace --stats of bench.py --steps 120): cache probe + plan, "
                                                 "pack of the live planes, head convolutions, FC layers, value head, softmax + gather + cache store. "
                                                 "library_kernels_*: kernels that are not this library's (torch elementwise / GEMM / copies), whole run "
                                                 "(set-up, preroll and the CPU-side bookkeeping included) divided by the number of steps"}
    def bench_line(name):
        p = os.path.join(out_dir, name)
        if not os.path.exists(p):
            return None
        try:
            return json.loads([l for l in open(p).read().splitlines() if l.startswith("{")][-1])
        except Exception:
            return None

    # both window passes in THIS run: they are compared with each other, not with what an earlier run left in the merged summary
    both_here = all(find(os.path.join(out_dir, k), "*counter_collection.csv") for k in ("pmc_fetch", "pmc_write"))
    if both_here:
        for key in ("k_bar", "d_bar", "warning"):
            summary.get("k_step", {}).get("window", {}).pop(key, None)
    for kind, counter in (("pmc_fetch", "FETCH_SIZE"), ("pmc_write", "WRITE_SIZE"), ("pmc_cfetch", "FETCH_SIZE"), ("pmc_cwrite", "WRITE_SIZE"),
                          ("pmc_hfetch", "FETCH_SIZE"), ("pmc_hwrite", "WRITE_SIZE")):   # h*: the evaluator's tail kernels (round 4)
        cc = find(os.path.join(out_dir, kind), "*counter_collection.csv")
        if not cc:
            continue
        acc = defaultdict(lambda: [0.0, 0])
        kstep = []   # (dispatch id, value) of every k_step launch of the pass, to cut out the timed window below
        for r in csv.DictReader(open(cc)):
            if r.get("Counter_Name") != counter:
                continue
            name = r.get("Kernel_Name", "")
            for k in ("k_step", "k_softmax_gather", "k_select", "k_expand_backup", "k_finish_move", "k_conv3x3", "k_head_conv1x1", "k_fc_f16", "k_fc_wide_f16",
                      "k_pack_live_planes", "k_cache_plan", "k_cache_probe", "k_value_out"):
                if k in name:
                    acc[k][0] += float(r.get("Counter_Value", 0) or 0)
                    acc[k][1] += 1
                    if k == "k_step":
                        kstep.append((int(float(r.get("Dispatch_Id", 0) or 0)), float(r.get("Counter_Value", 0) or 0)))
        for k, (tot, n) in acc.items():
            if n:
                summary.setdefault(k, {})[counter + "_raw_per_launch"] = tot / n
                summary[k][counter + "_launches"] = n
        # the k_step launches of the pass's TIMED WINDOW are its last (steps - move boundaries) ones: nothing launches k_step
        # after the window. Their counters and the pass's own k-bar / d-bar / algorithmic bytes describe the same trees.
        bl = bench_line(kind + "_bench.json")
        if bl and kstep and kind in ("pmc_fetch", "pmc_write"):
            n_win = int(bl["steps"]) - int(bl["move_boundary"]["in_window"])
            vals = [v for _, v in sorted(kstep, key=lambda t: t[0])][-n_win:]   # stable: file order if there is no dispatch id
            w = summary.setdefault("k_step", {}).setdefault("window", {})
            w[counter + "_raw_per_launch"] = sum(vals) / len(vals)
            w["launches"] = len(vals)
            rf = bl["roofline"]
            shape = {"k_bar": rf["k_bar"], "d_bar": rf["d_bar"], "algorithmic_bytes_per_launch": rf["algorithmic_bytes_per_launch"]}
            if "k_bar" in w and (abs(w["k_bar"] - shape["k_bar"]) > 1e-9 or abs(w["d_bar"] - shape["d_bar"]) > 1e-9):
                w["warning"] = f"the fetch and write passes saw different trees: {shape} vs k_bar {w['k_bar']} d_bar {w['d_bar']}"
            w.update(shape)
            w["bench_args"] = f"--steps {bl['steps']} --warmup {bl['warmup']} (window: {bl['config']['window']})"
    for sqdir in ("pmc_sq", "pmc_csq"):
      cc = find(os.path.join(out_dir, sqdir), "*counter_collection.csv")
      if cc:  # SQ counters per launch (quad-cycle units for the *_CYCLES / WAIT / ACTIVE counters, MI355X_MICROARCH.md)
          acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
          for r in csv.DictReader(open(cc)):
              name = r.get("Kernel_Name", "")
              for k in ("k_step", "k_softmax_gather", "k_finish_move", "k_harvest", "k_conv3x3"):
                  if k in name:
                      a = acc[k][r.get("Counter_Name")]
                      a[0] += float(r.get("Counter_Value", 0) or 0)
                      a[1] += 1
          for k, d in acc.items():
              sq = {c: v[0] / v[1] for c, v in d.items() if v[1]}
              wc = sq.get("SQ_WAVE_CYCLES")
              if wc:
                  sq["wait_any_frac"] = sq.get("SQ_WAIT_ANY", 0.0) / wc
                  sq["wait_inst_any_frac"] = sq.get("SQ_WAIT_INST_ANY", 0.0) / wc
                  sq["active_inst_any_frac"] = sq.get("SQ_ACTIVE_INST_ANY", 0.0) / wc
              summary.setdefault(k, {})["sq_per_launch"] = sq
    # matrix-pipe and LDS counters of the tower convolution (own pass: the SQ block has 8 counter slots)
    cc = find(os.path.join(out_dir, "pmc_cmfma"), "*counter_collection.csv")
    if cc:
        acc = defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(cc)):
            if "k_conv3x3" in r.get("Kernel_Name", ""):
                a = acc[r.get("Counter_Name")]
                a[0] += float(r.get("Counter_Value", 0) or 0)
                a[1] += 1
        m = {c: v[0] / v[1] for c, v in acc.items() if v[1]}
        if m.get("SQ_WAVE_CYCLES"):
            # SQ_VALU_MFMA_BUSY_CYCLES counts cycles (16 per v_mfma_f32_16x16x32_f16), SQ_WAVE_CYCLES quad-cycles summed over the
            # waves (MI355X_MICROARCH.md, price list); the kernel keeps two waves on every SIMD: the share of a SIMD's resident
            # time in which its matrix pipe works
            m["mfma_pipe_use_while_resident"] = m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * m["SQ_WAVE_CYCLES"] / 2.0)
        if m.get("SQ_LDS_IDX_ACTIVE"):
            m["lds_bank_conflict_over_active"] = m.get("SQ_LDS_BANK_CONFLICT", 0.0) / m["SQ_LDS_IDX_ACTIVE"]
        summary.setdefault("k_conv3x3", {})["mfma_lds_per_launch"] = m
    for k, d in summary.items():
        f = d.get("FETCH_SIZE_raw_per_launch")
        w = d.get("WRITE_SIZE_raw_per_launch")
        if f is not None and w is not None:
            # rocprofv3 FETCH_SIZE / WRITE_SIZE are in kilobytes; gfx950: double the fetch side
            d["hbm_bytes_per_launch"] = (2.0 * f + w) * 1024.0
            d["hbm_bytes_per_launch_uncorrected"] = (f + w) * 1024.0
        win = d.get("window")
        if win and "FETCH_SIZE_raw_per_launch" in win and "WRITE_SIZE_raw_per_launch" in win:
            win["hbm_bytes_per_launch"] = (2.0 * win["FETCH_SIZE_raw_per_launch"] + win["WRITE_SIZE_raw_per_launch"]) * 1024.0
            win["hbm_bytes_per_launch_uncorrected"] = (win["FETCH_SIZE_raw_per_launch"] + win["WRITE_SIZE_raw_per_launch"]) * 1024.0
            win["traffic_over_algorithmic"] = win["hbm_bytes_per_launch"] / win["algorithmic_bytes_per_launch"]
            win["traffic_over_algorithmic_uncorrected"] = win["hbm_bytes_per_launch_uncorrected"] / win["algorithmic_bytes_per_launch"]
            win["note"] = ("counters and algorithmic bytes of the SAME launches: the timed window of the PMC passes (real-net alignment). "
                           "fetch x2 = the gfx950 correction for wide coalesced reads (MI355X_MICROARCH.md, HBM); k_step's reads are mostly "
                           "16-B records and scattered 4-B words, so the truth lies between the two ratios")
    for name in ("trace_bench.json", "pmc_fetch_bench.json", "pmc_write_bench.json", "pmc_sq_bench.json", "pmc_cfetch_bench.json",
                 "pmc_cwrite_bench.json", "pmc_csq_bench.json"):
        p = os.path.join(out_dir, name)
        if os.path.exists(p):
            try:
                line = [l for l in open(p).read().splitlines() if l.startswith("{")][-1]
                summary.setdefault("_bench", {})[name] = json.loads(line)
            except Exception:
                pass
    with open(os.path.join(root, f"{tag}_summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    pm = {k: v for k, v in summary.items() if not k.startswith("_")}
    tb = summary.get("_bench", {}).get("trace_bench.json")
    if tb:  # bench.py replays these figures only for the workload they were measured on
        pm["workload"] = {"boards_per_gpu": tb["config"]["boards_per_gpu"], "sims_per_move": tb["config"]["sims_per_move"],
                          "evaluator": tb["config"]["evaluator"], "max_plies": tb["config"]["max_plies"]}
        # the code the profile was taken with (chinesechesszero_amd.build.code_hash(): kernels, C ABI, launch loop, evaluator): bench.py uses
        # the profile's k_step duration and counters only while it runs the same code
        pm["head"] = tb.get("roofline", {}).get("code_hash")
        heads = {name: b.get("roofline", {}).get("code_hash") for name, b in summary.get("_bench", {}).items()}
        if len(set(heads.values())) > 1:
            pm["head_warning"] = f"the passes were taken with different code: {heads}"
    pm["tag"] = tag
    pm["run"] = (f"profiles/run_profile.sh {tag}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / SQ_* in separate passes of bench.py --steps 24, "
                 "counter collection filtered per kernel (k_step: the bench's full command shape; k_conv3x3: short passes)")
    with open(os.path.join(root, "pmc_summary.json"), "w") as f:
        json.dump(pm, f, indent=1)
    print(json.dumps({k: v for k, v in summary.items() if not k.startswith("_")}, indent=1))


if __name__ == "__main__":
    main()
