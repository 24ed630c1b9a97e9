"""Host-side mirrors of the reference's containers and proposal styles that need no GPU: NamedTuple proposals fold into the
vector-of-Normals form, bundle_samples' containers, named parameters in traced closures.
Reference: src/AdvancedMH.jl:96-125 (bundle_samples), test/runtests.jl:112-201."""
import math

import numpy as np
import pytest

import mhx
import mhx.trace as T


def test_namedtuple_of_proposals_folds_into_a_vector_of_normals():
    spl = mhx.MetropolisHastings({"μ": mhx.StaticProposal(mhx.Normal(0, 1)), "σ": mhx.StaticProposal(mhx.Normal(0.5, 2.0))})
    assert isinstance(spl.proposal, mhx.StaticProposal) and spl.param_names == ["μ", "σ"]
    assert spl.proposal.proposal.dim == 2
    assert np.allclose(spl.proposal.proposal.mean, [0, 0.5]) and np.allclose(spl.proposal.proposal.vec, [1.0, 2.0])
    rw = mhx.MetropolisHastings({"a": mhx.RandomWalkProposal(mhx.Normal(0, 0.3)), "b": mhx.SymmetricRandomWalkProposal(mhx.Normal(0, 0.4))})
    assert isinstance(rw.proposal, mhx.RandomWalkProposal) and not rw.proposal.issymmetric and rw.param_names == ["a", "b"]
    with pytest.raises(mhx.ArgumentError, match="every entry"):
        mhx.MetropolisHastings({"a": mhx.RandomWalkProposal(mhx.Normal(0, 1)), "b": mhx.StaticProposal(mhx.Normal(0, 1))})
    with pytest.raises(mhx.ArgumentError, match="one scalar Normal"):
        mhx.MetropolisHastings({"a": mhx.StaticProposal(mhx.MvNormal(mhx.zeros(2), mhx.I))})


def test_bundle_samples_containers():
    rng = np.random.default_rng(0)
    value = rng.normal(size=(5, 3, 2))                     # 5 draws, 2 parameters + lp, 2 chains
    acc = rng.integers(0, 2, size=(5, 2)).astype(np.uint8)
    names = ["a", "b", "lp"]
    tr = mhx.bundle_samples(value, acc, names, mhx.Transition)
    assert len(tr) == 2 and len(tr[0]) == 5
    assert (tr[1][3].params == value[3, :2, 1]).all() and tr[1][3].lp == value[3, 2, 1] and tr[1][3].accepted == bool(acc[3, 1])
    nt = mhx.bundle_samples(value, acc, names, dict)
    assert tuple(nt[0][0].keys()) == ("a", "b", "lp") and nt[0][4]["b"] == value[4, 1, 0]       # test/runtests.jl:198-201
    sa = mhx.StructArray(value[:, :, :1], names)
    assert sa.keys() == ("a", "b", "lp") and sa.a.shape == (5,) and (sa.lp == value[:, 2, 0]).all() and len(sa) == 5
    assert mhx.StructArray(value, names).b.shape == (5, 2)


def test_named_parameters_in_a_traced_closure():
    """m3 = DensityModel(x -> logpdf(Normal(x.a, x.b), 1.0))  (test/runtests.jl:184)"""
    def m3(x):
        z = (1.0 - x.a) / x.b
        return T.where(x.b > 0, -0.5 * z * z - T.log(x.b) - 0.5 * math.log(2 * math.pi), -math.inf)

    model = mhx.DensityModel(m3, names=("a", "b"))
    assert model.dim == 2 and model.param_names == ["a", "b"]
    want = -0.5 * ((1.0 - 0.3) / 1.7) ** 2 - math.log(1.7) - 0.5 * math.log(2 * math.pi)
    assert abs(model.traced.evaluate([0.3, 1.7]) - want) < 1e-15
    by_key = T.trace(lambda x: x["a"] * x["b"] + x[0], 2, names=("a", "b"))
    assert by_key.evaluate([2.0, 5.0]) == 12.0
    with pytest.raises(T.TraceError, match="names"):
        T.trace(lambda x: x[0], 3, names=("a", "b"))


def test_mala_takes_the_reference_closure_form():
    """MALA(x -> MvNormal((σ² / 2) .* x, σ² * I))  (test/runtests.jl:291,352)"""
    s2 = 0.37
    spl = mhx.MALA(lambda g: mhx.MvNormal(0.5 * s2 * g, s2 * mhx.I))
    assert spl.resolve(5) == pytest.approx(s2, rel=1e-15) and mhx.MALA(s2).resolve(5) == s2
    diag = mhx.MALA(lambda g: mhx.MvNormal(0.5 * s2 * g, np.full(g.size, s2)))           # the same covariance written as a vector
    assert diag.resolve(3) == pytest.approx(s2, rel=1e-15)
    for wrong in (lambda g: mhx.MvNormal(g, mhx.I),                          # test/runtests.jl:43: not a Langevin step (mean g, not g / 2)
                  lambda g: mhx.MvNormal(0.5 * s2 * g + 1.0, s2 * mhx.I),   # shifted
                  lambda g: mhx.MvNormal(0.5 * s2 * g, np.linspace(1, 2, g.size)),   # anisotropic
                  lambda g: 3.0):
        with pytest.raises(mhx.ArgumentError, match="Langevin"):
            mhx.MALA(wrong).resolve(4)
