"""profiles/ must reproduce itself (Next 6): the counter summaries and traffic figures that DESIGN.md, README and
bench.py quote are regenerated here, on the CPU, from the rocprofv3 CSVs tracked beside them (filtered to the kernel by
scripts/pmc_filter.py), with the very scripts that made them, and compared with the tracked outputs."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = os.path.join(ROOT, "profiles")


def _run(script, *args):
    return subprocess.check_output([sys.executable, os.path.join("scripts", script)] + list(args), cwd=ROOT).decode()


@pytest.mark.parametrize("summary,kernel,csvs", [
    ("r05_pmc_summary_k_sweep64_mfma2.txt", "k_sweep64_mfma2",
     ["profiles/r05_pmc/pmc1_k_sweep64_mfma2.csv", "profiles/r05_pmc/pmc2_k_sweep64_mfma2.csv"]),
    ("r05_pmc_summary_decisions_alone_cfg3.txt", "k_block_chain2_t",
     ["profiles/r05_pmc/decision_pmc_a.csv", "profiles/r05_pmc/decision_pmc_b.csv"]),
    ("r04_pmc_summary_fused_k_sweep64_mfma2.txt", "k_sweep64_mfma2",
     ["profiles/r04_pmc/pmc1_f64.csv", "profiles/r04_pmc/pmc2_f64.csv"]),   # (round 4 tracked CSVs of another run: replaced)
    ("r04_pmc_summary_fused_k_sweep32_pull.txt", "k_sweep32_pull", ["profiles/r04_pmc/pmc1_f32.csv"]),
    ("r03_sweep_cfg4_pmc_summary.txt", None, None),
])
def test_counter_summaries_reproduce(summary, kernel, csvs):
    path = os.path.join(P, summary)
    if not os.path.exists(path):
        pytest.skip(summary + " is not tracked (yet)")
    tracked = open(path).read()
    if csvs is None:   # round 3's summary names its own inputs in its '# <csv>:' header lines
        csvs = [ln[2:].split(":")[0] for ln in tracked.splitlines() if ln.startswith("# profiles/")]
        kernel = "k_sweep32_pull"
        if not csvs:
            pytest.skip("the summary does not name tracked inputs")
    for c in csvs:
        assert os.path.exists(os.path.join(ROOT, c)), c
    assert _run("pmc_summary.py", kernel, *csvs) == tracked


@pytest.mark.parametrize("name,args", [
    ("r05_traffic_cfg4_block64.json", ["profiles/r05_pmc/fetch_k_sweep64_mfma2.csv", "profiles/r05_pmc/write_k_sweep64_mfma2.csv",
                                       "k_sweep64_mfma2", "32768", "16384", "cfg4", "64", "256"]),
])
def test_traffic_figures_reproduce(name, args):
    path = os.path.join(P, name)
    if not os.path.exists(path):
        pytest.skip(name + " is not tracked (yet)")
    tracked = json.load(open(path))
    again = json.loads(_run("pmc_traffic.py", *args))
    again["date"] = tracked["date"]   # (the day the file was made)
    assert again == tracked
    assert 1.0 <= tracked["ratio_traffic_over_one_pass"] < 1.06, tracked   # what DESIGN.md says of the sweep's traffic


def test_documents_quote_the_tracked_traffic_ratio():
    """DESIGN.md / README.md state the MFMA sweep's traffic as a multiple of one pass: the number must be the tracked one."""
    src = None
    for name in ("r05_traffic_cfg4_block64.json", "r04_traffic_cfg4_fused_block64.json"):
        if os.path.exists(os.path.join(P, name)):
            src = json.load(open(os.path.join(P, name)))
            break
    assert src is not None
    want = "%.3f" % src["ratio_traffic_over_one_pass"]
    for doc in ("DESIGN.md", "README.md"):
        text = open(os.path.join(ROOT, doc)).read()
        assert want in text, "%s does not quote the tracked traffic ratio %s of k_sweep64_mfma2" % (doc, want)
