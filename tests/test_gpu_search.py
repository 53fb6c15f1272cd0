"""End to end through OUR host (reader -> cut -> device encode -> batched SA -> treestack mirror ->
tree writer): the reference's black-box known answers must come out (tests/golden/ref_tests.json).
The scorer is bit-exact (other tests); this checks the search mirror drives it to the same optima."""
import json
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

GOLD = Path(__file__).resolve().parent / "golden"
WANT = {c["infile"]: c["expect"] for c in json.loads((GOLD / "ref_tests.json").read_text())["cases"]}


@pytest.mark.parametrize("phy,score,topologies,on_device", [
    ("test_treelength_5_thread_2.phy", 1846, None, 0),
    ("test_treelength_6_thread_2.phy", 1628, 1, 0),
    ("test_treelength_6_thread_2.phy", 1628, 1, 1),
    ("test_treelength_7_thread_2.phy", 1006, 1, 1),
    ("test_treelength_6_thread_3.phy", 297, None, 2),
], ids=["5t2-host", "6t2-host", "6t2-device", "7t2-device", "6t3-auto"])
def test_search_reaches_reference_optimum(tmp_path, phy, score, topologies, on_device):
    from lvb_amd import search
    out = tmp_path / "outtree"
    res = search.run(str(GOLD / "ref_tests" / phy), seed=509739986, algorithm=1, batch=64, out=str(out),
                     max_seconds=60, verbose=False, device_proposals=on_device)
    assert res["best_length"] == score, res
    if topologies is not None:
        assert res["topologies"] == topologies
    lines = out.read_text().splitlines()
    assert len(lines) == min(res["topologies"], 1024) and all(l.startswith("(") and l.endswith(");") for l in lines)
