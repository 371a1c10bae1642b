"""The committed rocprofv3 passes under profiles/ parse with the tools that produced the published summaries, and the
summaries agree with them (keeps `roofline.traffic` in bench.py reproducible from the raw counter files)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")
DOMINANT = "void fcn8s::gemm_glds_kernel<128, 128, 2, 2, 3, false>(fcn8s::IgemmArgs)"


def test_pmc_traffic_reproducible(tmp_path):
    out = tmp_path / "t.json"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), os.path.join(P, "r06_pmc_fetch_counter_collection.csv"),
                    os.path.join(P, "r06_pmc_write_counter_collection.csv"), str(out), "test"], check=True, capture_output=True)
    got = json.load(open(out))["kernels"][DOMINANT]
    pub = json.load(open(os.path.join(P, "pmc_traffic.json")))["kernels"][DOMINANT]
    for k in ("launches", "fetch_mb_per_launch", "write_mb_per_launch", "hbm_mb_per_launch", "fetch_mb_per_launch_uncorrected"):
        assert got[k] == pub[k], k
    assert abs(got["fetch_mb_per_launch"] - 2 * got["fetch_mb_per_launch_uncorrected"]) < 0.01     # the gfx950 FETCH_SIZE correction
    bench = json.load(open(os.path.join(P, "r06_bench_train_bs16.json")))
    assert bench["roofline"]["kernel"] in DOMINANT.replace("void fcn8s::", "")
    assert abs(bench["roofline"]["frac"] - bench["roofline"]["achieved"] / bench["roofline"]["peak"]) < 1e-3
    if bench["roofline"]["traffic"] is not None:
        src = (bench["roofline"].get("traffic_detail") or {}).get("traffic_source", "")
        if src.startswith("live"):         # re-measured by the bench run itself (two PMC passes of its own): the same figure to well under a percent
            assert abs(bench["roofline"]["traffic"] - pub["hbm_mb_per_launch"] * 1e6) < 5e-3 * pub["hbm_mb_per_launch"] * 1e6
        else:                              # the line was printed with this very pmc_traffic.json in place
            assert bench["roofline"]["traffic"] == round(pub["hbm_mb_per_launch"] * 1e6)


def test_clock_summary_reproducible(tmp_path):
    out = tmp_path / "c.json"
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_clock_summary.py"), os.path.join(P, "r06_pmc_clock_counter_collection.csv"), str(out)],
                   check=True, capture_output=True)
    got = json.load(open(out))[DOMINANT]
    pub = json.load(open(os.path.join(P, "r06_pmc_clock.json")))[DOMINANT]
    assert got == pub
    assert 1.5 < got["effective_clock_ghz"] <= 2.45 and 0.5 < got["mfma_pipe_busy"] <= 1.0


def test_launch_gap_report_on_a_synthetic_trace(tmp_path):
    """tools/launch_gap_report.py: 15 passes of three kernels each, 100 us long, 2 us apart, 10 us between passes."""
    import csv
    path = tmp_path / "t_kernel_trace.csv"
    t = 1_000_000
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kind", "Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        for _ in range(15):
            for name in ("fcn8s::preprocess_kernel(void const*)", "a", "b"):
                w.writerow(["KERNEL_DISPATCH", name, t, t + 100_000])
                t += 102_000
            t += 8_000
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "launch_gap_report.py"), str(path)], check=True, capture_output=True, text=True).stdout
    assert "wall span per pass           314.0 us" in out and "a kernel is running          300.0 us" in out, out
    assert "idle between its kernels       4.0 us" in out and "3 launches, 2.00 us per gap" in out and "idle before the next pass     10.0 us" in out, out


def test_roofline_table_tool_reads_the_committed_bench_line():
    """tools/roofline_table.py: every kernel group of the committed bench line of the current round against the roof that bounds it; the table committed
    under profiles/ is what the tool prints for that line."""
    import subprocess, sys
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "roofline_table.py"), os.path.join("profiles", "r06_bench_train_bs16.json")],
                         capture_output=True, text=True, check=True, cwd=ROOT).stdout
    assert out == open(os.path.join(ROOT, "profiles", "r06_roofline_table.md")).read()
    rows = [l.split("|") for l in out.splitlines() if l.startswith("| ") and "kernel group" not in l]
    assert len(rows) >= 15
    by = {r[1].strip(): r for r in rows}
    assert by["wino_transform"][4].strip() == "HBM" and 0.4 < float(by["wino_transform"][7]) < 1.0
    assert by["wino_gemm_fc6_fwd"][4].strip() == "f32 MFMA" and 0.5 < float(by["wino_gemm_fc6_fwd"][7]) <= 1.0
    shares = sum(float(r[3].replace("%", "")) for r in rows)
    assert 99.0 < shares < 101.0
