"""The program specialiser (csrc/gm_spec.hip) without a GPU: the source it generates for real programs
-- taken from random views through the product's own compiler -- must compile for gfx950 (hiprtc
cross-compiles), name every micro-op, and turn table ops into literals where it promises to."""
import numpy as np
import pytest

from fuzz_util import REQ, random_view
from dask_geomodeling_amd.hip import program as P


def _collect_programs(monkeypatch, seeds, depth=4):
    from oracle import vm_numpy
    seen = {}

    def launch(compiled, tiles, shape):
        seen.setdefault(bytes(compiled.program)[:64] + bytes(compiled.program)[-64:] + bytes([compiled.n_steps]), compiled)
        return vm_numpy.launch(compiled, tiles, shape)

    monkeypatch.setattr(P, "launch", launch)
    monkeypatch.setattr(P, "launch_batch", lambda compiled, tiles, shape, outs=None: [launch(compiled, t, shape) for t in tiles])
    for seed in seeds:
        view = random_view(np.random.default_rng(70_000 + seed), "s%d" % seed, depth=depth)
        try:
            with np.errstate(all="ignore"):
                view.get_data(**REQ)
        except NotImplementedError:
            pass
    return list(seen.values())


@pytest.mark.skipif(not P.spec_available(), reason="hiprtc is not installed")
def test_generated_kernels_compile_for_gfx950(cpu_vm, monkeypatch):
    programs = _collect_programs(monkeypatch, range(40))
    assert len(programs) >= 20
    classes = {c.reg_class for c in programs}
    assert len(classes) >= 3  # float32, float64 and an integer class at least
    # the longest programs of every register class, both launch shapes
    picked = []
    for cls in sorted(classes):
        of_class = sorted((c for c in programs if c.reg_class == cls), key=lambda c: -c.n_steps)
        picked += of_class[:4]
    for k, comp in enumerate(picked):
        src = P.spec_source(comp, strided=bool(k % 2), group=1 + k % 2)
        assert "void gm_spec_" in src and src.count("m.") >= 1
        assert P.spec_compile_check(comp, strided=bool(k % 2), group=1 + k % 2) > 1000


@pytest.mark.skipif(not P.spec_available(), reason="hiprtc is not installed")
def test_classify_reclassify_mask_become_one_literal_statement():
    """Mask(Reclassify(Classify(x))) -- BASELINE configs[4]'s tail: DIGITIZE + LOOKUP (Mask folded into the
    table by the host) is tabulated over the edge counts; no table load, no LDS, no span loop left."""
    from dask_geomodeling_amd.raster import devops, tables
    ndv = np.float32(np.finfo("f4").max)
    kw = {"dtype": "float32", "fillvalue": float(ndv)}
    a, b = P.Leaf(0, np.dtype("f4"), ndv), P.Leaf(1, np.dtype("f4"), ndv)
    x = devops.math_node("add", kw, [a, b])
    x = tables.classify_node(x, [-50.0, -10.0, 0.0, 10.0, 50.0], False)
    x = tables.reclassify_node(x, {"dtype": "<i8", "fillvalue": int(np.iinfo("i8").max), "select": True,
                                   "data": [[0, 10], [1, 11], [3, 13], [5, 15]]})
    comp = P.compile_expr(devops.mask_node(x, 7), 2)
    src = P.spec_source(comp)
    assert "op_digimap_lit" in src and "op_lookup" not in src.replace("op_digimap_lit", "")
    assert "// flavor=0 lds=0" in src
    # select=True: counts 2 and 4 are unmapped -> fill -> Mask's fill (0); the others -> 7.  Equal neighbours
    # are merged, so 4 of the 5 edges remain (c = 0|1 -> 7, 2 -> 0, 3 -> 7, 4 -> 0, 5 -> 7)
    line = [ln for ln in src.splitlines() if "op_digimap_lit" in ln][0]
    edges = line.split("Bits<")[1].split(">")[0].split(",")
    assert len(edges) == 4
    assert P.spec_compile_check(comp) > 1000
