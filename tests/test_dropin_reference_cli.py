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


@pytest.mark.skipif(not os.path.isdir(REF), reason="reference checkout not present")
def test_reference_probe_designer_reaches_gpu_clustering(monkeypatch):
    """--cluster-and-design-separately: the reference's ProbeDesigner calls
    catch.utils.cluster.cluster_with_minhash_signatures; with catch_amd's
    module registered there it receives the sequences (and, with a GPU, would
    return the same clusters -- tests/test_gpu_parity.py checks that against
    recorded reference output)."""
    from catch_amd import _lib, engine
    from catch_amd.utils import cluster as gpu_cluster
    monkeypatch.syspath_prepend(REF)
    monkeypatch.setattr(sys, "dont_write_bytecode", True)
    import catch.utils
    from catch import genome
    from catch.filter import duplicate_filter, probe_designer
    monkeypatch.setitem(sys.modules, "catch.utils.cluster", gpu_cluster)
    monkeypatch.setattr(catch.utils, "cluster", gpu_cluster, raising=False)
    monkeypatch.setattr(probe_designer, "cluster", gpu_cluster)
    seen = {}
    orig = gpu_cluster.cluster_with_minhash_signatures

    def spy(seqs, **kw):
        seen["n"] = len(seqs)
        seen["kw"] = kw
        return orig(seqs, **kw)
    monkeypatch.setattr(gpu_cluster, "cluster_with_minhash_signatures", spy)
    gs = [[genome.Genome.from_one_seq("ATTA" * 500), genome.Genome.from_one_seq("CGGC" * 500)]]
    df = duplicate_filter.DuplicateFilter()
    pd = probe_designer.ProbeDesigner(gs, [df], probe_length=100, probe_stride=50,
                                      cluster_threshold=0.1, cluster_merge_after=df,
                                      cluster_method="simple", cluster_fragment_length=500)
    if engine.device_count() > 0:
        assert len(pd._cluster_genomes()) == 2
    else:
        with pytest.raises(_lib.CatchHipError):
            pd._cluster_genomes()
    assert seen["n"] == 8 and seen["kw"] == dict(threshold=0.1, cluster_method="simple")
