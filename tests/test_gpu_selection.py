"""GPU (-m gpu): the kernel-selection tables as a GATE.  tools/selection_check.py re-measures, for a probe on both
sides of every table row (gemm_splitk_plan, gemm_pp128_wins, deep_plan_auto, wo_skinny_pick, wo_wide_plan, gemm_takes_skinny), the
automatic choice against each alternative the row chose between (forced through the debug knobs, device-paced HIP graphs, cold
weights where the row was fitted cold) and flags a probe whose automatic choice is more than 10 % behind its best alternative.
Here: the quick set (one probe per row); a flagged probe is measured a second time in a fresh process and fails the test only if it
is flagged again (box-to-box spread of single launches on this pool is +-4 %).  Skipped where the CU count is not the 256 the tables
were fitted on."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "selection_check.py"), "--quick", "--tolerance", "0.10"],
                       cwd=ROOT, capture_output=True, text=True, timeout=1500, env=dict(os.environ, MIXQ_DEBUG_KNOBS="1"))
    flagged = set(re.findall(r"^\s{3}(\S+(?: \(cold\))?) (\d+x\d+x\d+): \+", r.stdout, flags=re.M))
    assert r.returncode in (0, 1), (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
    assert "probe(s) more than" in r.stdout, r.stdout[-2000:]
    return r, flagged


def test_automatic_selection_is_within_10_percent_of_the_best_alternative_on_every_table_row():
    if torch.cuda.get_device_properties(0).multi_processor_count != 256:
        pytest.skip("the selection tables are fitted on a 256-CU part")
    r, flagged = _run()
    if flagged:
        r2, flagged2 = _run()
        both = flagged & flagged2
        assert not both, f"flagged twice: {sorted(both)}\n{r.stdout[-3000:]}\n{r2.stdout[-3000:]}"
