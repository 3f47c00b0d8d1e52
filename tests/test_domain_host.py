"""Host-side logic of Domain / State (no kernels): flattening order, multigrid level shapes,
geometry -- reference tests/test_domain.py:12-59 restated, plus geometry vs the oracle.
Runs on CPU tensors (`ModRocm(device='cpu')` exists for exactly this plumbing check)."""

from copy import deepcopy

import numpy as np
import pytest

import odil_amd as odil
from oracle import odil_np as onp


@pytest.fixture(scope="module")
def mod():
    return odil.ModRocm(device="cpu")


@pytest.mark.parametrize("case", ["pack", "arrays"])
@pytest.mark.parametrize("dim", [1, 2])
def test_pack_unpack_equals_arrays_equals_direct(mod, case, dim):
    cshape = tuple((1 + np.arange(dim)) * 2)
    dimnames = ["x", "y", "z", "w"][:dim]
    domain = odil.Domain(cshape=cshape, dimnames=dimnames, multigrid=1, mg_convert_all=False, mod=mod,
                         dtype=np.float64)
    state = odil.State(
        fields={
            "field": np.random.rand(*cshape),
            "mgfield": domain.regular_to_multigrid(np.random.rand(*cshape)),
            "net": domain.make_neural_net([3, 3]),
            "array": [1, 2, 3],
        }
    )
    state = domain.init_state(state)
    state2 = deepcopy(state)
    upd = lambda u: u + 1
    if case == "pack":
        packed = upd(domain.pack_state(state))
        domain.unpack_state(packed, state)
    else:
        arrays = [upd(a) for a in domain.arrays_from_state(state)]
        domain.arrays_to_state(arrays, state)
    for f in state2.fields.values():
        if isinstance(f, odil.core.Field):
            f.array = upd(f.array)
        elif isinstance(f, odil.core.MultigridField):
            for t in f.terms:
                t.array = upd(t.array)
        elif isinstance(f, odil.core.NeuralNet):
            for i in range(len(f.weights)):
                f.weights[i] = upd(f.weights[i])
                f.biases[i] = upd(f.biases[i])
        elif isinstance(f, odil.core.Array):
            f.array = upd(f.array)
    assert float((domain.pack_state(state) - domain.pack_state(state2)).abs().max()) == 0.0
    # flattening order: dict order; MG terms fine -> coarse; weights then biases (reference core.py:361-383)
    kinds = [tuple(a.shape) for a in domain.arrays_from_state(state)]
    assert kinds[0] == cshape and kinds[-1] == (3,) and kinds[-2] == (3,) and kinds[-3] == (3, 3)


def test_geometry_matches_oracle(mod):
    domain = odil.Domain(cshape=(4, 6), dimnames=["x", "y"], lower=(0, -1), upper=(2, 1), dtype=np.float64, mod=mod)
    for loc in ["cc", "nn", "cn"]:
        got = [x.numpy() for x in domain.points(loc=loc)]
        xs = [onp.points_1d(lo, up, n, l, np.float64) for lo, up, n, l in zip((0, -1), (2, 1), (4, 6), loc)]
        want = np.meshgrid(*xs, indexing="ij")
        for a, b in zip(got, want):
            assert np.array_equal(a, b)
    assert domain.step() == (0.5, 1.0 / 3)
    assert domain.size(loc="nc") == [5, 6] and domain.get_field_shape("cn") == (4, 7)
    assert np.array_equal(domain.indices("y").numpy(), np.meshgrid(np.arange(4), np.arange(6), indexing="ij")[1])
    d = odil.Domain(cshape=(256,), multigrid=True, dtype=np.float64, mod=mod)
    assert d.mg_cshapes == onp.mg_cshapes((256,)) and d.mg_nlvl == 8
    d = odil.Domain(cshape=(8, 16), multigrid=True, dtype=np.float64, mod=mod, mg_axes=[True, False])
    assert d.mg_cshapes == onp.mg_cshapes((8, 16), [True, False])
    with pytest.raises(ValueError):
        odil.Domain(cshape=(6,), multigrid=True, dtype=np.float64, mod=mod)  # 6 -> 3 -> 1 does not halve


def test_compute_on_cpu_tensors_is_refused(mod):
    domain = odil.Domain(cshape=(8,), multigrid=True, dtype=np.float64, mod=mod)
    state = domain.init_state(odil.State(fields={"u": None}))
    problem = odil.Problem(lambda ctx: [ctx.field("u")], domain)
    with pytest.raises(odil._lib.OdilHipError):
        problem.eval_loss_grad(state)
