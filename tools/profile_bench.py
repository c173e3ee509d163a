#!/usr/bin/env python3
"""rocprofv3 evidence for bench.py's roofline objects, taken for EXACTLY the kernels a plain `python bench.py` runs.

    python3 tools/profile_bench.py [round-tag]          (on the GPU box; writes gpurun_out/profile_bench/, copy with collect_profiles.py)

For Q6 (the headline) and Q1:
  1. a plain, unprofiled run of bench.py (default --jit tune): its `roofline.kernel` names the form the tuner settled on
     ("k_mscan_specialised<4,3,vec,global,late2>_grid2048": 3 row pairs per lane, two filter columns read with the tile);
  2. the same command with VDL_JIT_PIN set to that form (the tuner then builds and times that one candidate only) under
     `rocprofv3 --kernel-trace --stats` -> <q>_kernel_stats.csv, and in a SEPARATE pass under `--pmc FETCH_SIZE --kernel-trace`
     (never combined with other trace domains) -> <q>_fetch_pmc.csv;
  3. the read-everything kernel (--jit off: k_scan / precompiled k_mscan) the same way.
traffic.json maps bench.py's kernel label -> {rows, query, rocprof kernel name, FETCH_SIZE mean, hbm_bytes_per_launch =
FETCH_SIZE x 1024 x 2}; the x2 is the gfx950 correction, which tools/ubench/fetch_calib shows to hold for masked loads too
(one TCC_EA0_RDREQ per 128-byte line touched, tallied at 64 B) -- its table is taken here as well (fetch_calib.txt).
Every rocprofv3 call has `python3 bench.py ...` directly behind `--` (no shell, no env wrapper: gpurun's rule)."""
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out", "profile_bench")
BENCH = os.path.join(ROOT, "bench.py")
os.makedirs(OUT, exist_ok=True)
os.chdir("/tmp")
os.environ["TMPDIR"] = "/tmp"


def sh(cmd, out, env=None):
    print("+", " ".join(cmd), flush=True)
    e = dict(os.environ)
    e.update(env or {})
    with open(out, "w") as f, open(out + ".err", "w") as g:
        rc = subprocess.call(cmd, stdout=f, stderr=g, env=e)
    if rc != 0:
        print(open(out + ".err").read()[-3000:])
        raise SystemExit("failed (%d): %s" % (rc, " ".join(cmd)))


def last_json(path):
    return json.loads([l for l in open(path).read().splitlines() if l.startswith("{")][-1])


def pin_of(label):
    """'k_mscan_specialised<4,3,vec,global,late2>_grid2048' -> 'u=3,late=2'; precompiled kernels -> None"""
    m = re.match(r"k_mscan_specialised<\d+,(\d+),[^>]*>", label)
    if not m:
        return None
    if ",queue>" in label:
        return "u=%s,late=3" % m.group(1)
    late = re.search(r",late(\d?)>", label)
    return "u=%s,late=%s" % (m.group(1), (late.group(1) or "1") if late else "0")


def one(tag):
    return sorted(glob.glob(os.path.join(OUT, tag, "*", "*")))


def scan_rows(pmc_csv, stats_names):
    """mean FETCH_SIZE per dispatch of the scan kernels in a counter_collection.csv; refuses a kernel the stats pass never saw"""
    per = {}
    for r in csv.DictReader(open(pmc_csv)):
        n = r["Kernel_Name"]
        if r["Counter_Name"] != "FETCH_SIZE" or not ("k_scan<" in n or "vdl_jit_mscan" in n or "k_mscan<" in n):
            continue
        per.setdefault(n, {}).setdefault(r["Dispatch_Id"], 0.0)
        per[n][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for n in per:
        if n not in stats_names:
            raise SystemExit("kernel %r of the PMC pass does not occur in the matching kernel-stats pass: the two passes ran different kernels" % n)
    return {n: (sum(d.values()) / len(d), len(d)) for n, d in per.items()}


traffic = {}
fast = ["--no-cpu-baseline", "--no-secondary"]
for query in ("q6", "q1"):
    q = ["--query", query] if query != "q6" else []
    plain = os.path.join(OUT, query + "_plain.json")
    sh(["python3", BENCH] + q + ["--steps", "20", "--warmup", "3"] + fast, plain)
    label = last_json(plain)["roofline"]["kernel"]
    pin = pin_of(label)
    print(query, "plain run chose", label, "-> pin", pin, flush=True)
    runs = [("tuned", ["--jit", "tune"], {"VDL_JIT_PIN": pin} if pin else {}, label)] if pin else []
    # the staged form's sibling (one / two filter columns read with the tile: within a per cent or two of each other, so either
    # may be what another box's tuner keeps): profiled too, so that traffic.json has an entry for whichever a run reports
    if pin and "late=" in pin and not pin.endswith("late=0"):
        other = pin[:-1] + ("1" if pin.endswith("2") else "2")
        runs.append(("sibling", ["--jit", "tune"], {"VDL_JIT_PIN": other}, None))
    runs.append(("everything", ["--jit", "off"], {}, None))
    for name, jit, env, want in runs:
        tag = "%s_%s" % (query, name)
        sh(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", os.path.join(OUT, tag), "--", "python3", BENCH] + q + jit +
           ["--steps", "20", "--warmup", "3"] + fast, os.path.join(OUT, tag + "_bench.json"), env)
        sh(["rocprofv3", "--pmc", "FETCH_SIZE", "--kernel-trace", "--output-format", "csv", "-d", os.path.join(OUT, tag + "_fetch"), "--", "python3", BENCH] + q + jit +
           ["--steps", "3", "--warmup", "1", "--no-verify", "--latency-steps", "0"] + fast, os.path.join(OUT, tag + "_fetch.json"), env)
        b = last_json(os.path.join(OUT, tag + "_bench.json"))
        got = b["roofline"]["kernel"]
        if want and got != want:
            raise SystemExit("the pinned run executed %s, the plain run %s" % (got, want))
        stats = next(f for f in one(tag) if f.endswith("kernel_stats.csv"))
        os.replace(stats, os.path.join(OUT, tag + "_kernel_stats.csv"))
        names = {r["Name"]: r for r in csv.DictReader(open(os.path.join(OUT, tag + "_kernel_stats.csv")))}
        pmc = next(f for f in one(tag + "_fetch") if f.endswith("counter_collection.csv"))
        rows = scan_rows(pmc, names)
        # the timed kernel = the scan kernel with the most launches in the stats pass
        kname = max(rows, key=lambda n: int(names[n]["Calls"]))
        fetch, nd = rows[kname]
        traffic[got] = {"query": query, "rows": b["config"]["rows_per_gpu"], "rocprof_kernel": kname, "rocprof_avg_ns": float(names[kname]["AverageNs"]),
                        "rocprof_calls": int(names[kname]["Calls"]), "FETCH_SIZE_KiB_mean": fetch, "pmc_dispatches": nd,
                        "hbm_bytes_per_launch": int(fetch * 1024 * 2),
                        "correction": "x2: one TCC_EA0_RDREQ per 128-byte line, tallied at 64 B (MI355X_MICROARCH.md HBM; holds for masked loads: fetch_calib.txt)",
                        "bytes_moved_per_launch_counted": b["roofline"]["bytes_moved_per_launch"], "algorithmic_bytes_per_launch": b["roofline"]["algorithmic_bytes_per_launch"],
                        "bench_kernel_us": b["roofline"]["kernel_us"],
                        # both fractions of the 8 TB/s peak for this kernel (rocprof's average duration): on the bytes it MOVED (FETCH x 2) and
                        # on SURVEY.md 8(d)'s algorithmic bytes -- equal for a kernel that reads everything, the latter above 1 for a staged one
                        "frac_of_peak_on_bytes_moved": fetch * 1024 * 2 / (float(names[kname]["AverageNs"]) * 1e-9) / 8e12,
                        "frac_of_peak_on_algorithmic_bytes": b["roofline"]["algorithmic_bytes_per_launch"] / (float(names[kname]["AverageNs"]) * 1e-9) / 8e12,
                        "source": "rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 bench.py %s%s --steps 3 --warmup 1%s (tools/profile_bench.py)" %
                                  (" ".join(q + jit), "", " [VDL_JIT_PIN=%s]" % env["VDL_JIT_PIN"] if env else "")}
        # keep the scan kernels' PMC rows
        keep = [r for r in csv.DictReader(open(pmc)) if r["Kernel_Name"] in rows]
        with open(os.path.join(OUT, tag + "_fetch_pmc.csv"), "w") as f:
            w = csv.DictWriter(f, fieldnames=list(keep[0].keys()))
            w.writeheader()
            w.writerows(keep)
        t = traffic[got]
        print("  %s: %s  rocprof avg %.1f us (%d calls), events %.1f us; FETCH x2 = %.4g B, counted %.4g B (ratio %.4f), algorithmic %.4g B; frac of peak on FETCH x2: %.3f" %
              (tag, kname, t["rocprof_avg_ns"] / 1e3, t["rocprof_calls"], t["bench_kernel_us"], t["hbm_bytes_per_launch"], t["bytes_moved_per_launch_counted"],
               t["hbm_bytes_per_launch"] / max(t["bytes_moved_per_launch_counted"], 1), t["algorithmic_bytes_per_launch"],
               t["hbm_bytes_per_launch"] / (t["rocprof_avg_ns"] * 1e-9) / 8e12), flush=True)
json.dump(traffic, open(os.path.join(OUT, "traffic.json"), "w"), indent=1)

# the plain command under the kernel trace (default --jit tune, secondary measurements included): the tuner's candidates show up
# as kernels of their own, the winner's average is the one bench.py reports
sh(["rocprofv3", "--kernel-trace", "--stats", "--output-format", "csv", "-d", os.path.join(OUT, "default"), "--", "python3", BENCH, "--steps", "20", "--warmup", "3",
    "--no-cpu-baseline"], os.path.join(OUT, "default_bench.json"))
os.replace(next(f for f in one("default") if f.endswith("kernel_stats.csv")), os.path.join(OUT, "default_kernel_stats.csv"))

# the calibration the x2 rests on
calib = os.path.join(ROOT, "tools", "ubench", "fetch_calib")
if os.path.exists(calib):
    sh([calib, "2", "3"], os.path.join(OUT, "fetch_calib_plain.txt"))
    sh(["rocprofv3", "--pmc", "FETCH_SIZE", "--kernel-trace", "--output-format", "csv", "-d", os.path.join(OUT, "calib_fetch"), "--", calib, "2", "3"], os.path.join(OUT, "fetch_calib_under_pmc.txt"))
    sh(["rocprofv3", "--pmc", "TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "--kernel-trace", "--output-format", "csv", "-d", os.path.join(OUT, "calib_rdreq"), "--", calib, "2", "3"],
       os.path.join(OUT, "fetch_calib_under_pmc2.txt"))
    with open(os.path.join(OUT, "fetch_calib.txt"), "w") as f:
        f.write("== plain run (times)\n" + open(os.path.join(OUT, "fetch_calib_plain.txt")).read())
        for tag, table in (("calib_fetch", "fetch_calib_under_pmc.txt"), ("calib_rdreq", "fetch_calib_under_pmc2.txt")):
            pmc = next(x for x in one(tag) if x.endswith("counter_collection.csv"))
            f.write("\n== counters (%s)\n" % tag)
            f.write(subprocess.check_output([sys.executable, os.path.join(ROOT, "tools", "ubench", "fetch_calib_join.py"), os.path.join(OUT, table), pmc, "3"], text=True))
print(json.dumps({k: (v["rocprof_kernel"], v["hbm_bytes_per_launch"], v["bytes_moved_per_launch_counted"]) for k, v in traffic.items()}, indent=1))
