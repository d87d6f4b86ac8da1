"""Drop-in boundary: the UNMODIFIED reference CLI (bin/design.py) picks up the
GPU SetCoverFilter when it is registered at the reference's import path
(INTEGRATION.md §3).  Needs the reference checkout, which exists only in the
authoring container; there is no GPU there, so the run must reach our filter
and stop at device creation (with a GPU it would complete)."""
import importlib
import os
import sys

import pytest

REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
def test_reference_design_cli_reaches_gpu_filter(tmp_path, monkeypatch):
    from catch_amd import _lib, engine
    from catch_amd.filter import set_cover_filter as gpu_scf
    monkeypatch.syspath_prepend(os.path.join(REF, "bin"))
    monkeypatch.syspath_prepend(REF)
    monkeypatch.setattr(sys, "dont_write_bytecode", True)
    import catch.filter
    monkeypatch.setitem(sys.modules, "catch.filter.set_cover_filter", gpu_scf)
    monkeypatch.setattr(catch.filter, "set_cover_filter", gpu_scf, raising=False)
    sys.modules.pop("design", None)
    design = importlib.import_module("design")
    assert design.set_cover_filter is gpu_scf

    from catch_amd.utils import synthetic
    import numpy as np
    rng = np.random.Generator(np.random.PCG64(1))
    g = synthetic.make_species(rng, [1500], 2, 1, 0.0, 0.02, with_n=False)
    fa = tmp_path / "t.fasta"
    fa.write_text("".join(">g%d\n%s\n" % (i, x[0]) for i, x in enumerate(g)))
    out = tmp_path / "probes.fasta"
    seen = {}
    orig = gpu_scf.SetCoverFilter._filter

    def spy(self, input, target_genomes_grouped):
        seen["groups"] = [len(x) for x in input]
        seen["genomes"] = [len(x) for x in target_genomes_grouped]
        seen["params"] = (self.mismatches, self.lcf_thres, self.cover_extension)
        return orig(self, input, target_genomes_grouped)

    monkeypatch.setattr(gpu_scf.SetCoverFilter, "_filter", spy)
    monkeypatch.setattr(sys, "argv", ["design.py", str(fa), "-pl", "75", "-m", "2",
                                      "-e", "50", "-o", str(out)])
    args = design.init_and_parse_args("basic")
    if engine.device_count() > 0:
        design.main(args)
        assert out.exists()
    else:
        with pytest.raises(_lib.CatchHipError):
            design.main(args)
    assert seen["genomes"] == [2] and seen["groups"][0] > 20
    assert seen["params"] == (2, 75, 50)
    sys.modules.pop("design", None)
