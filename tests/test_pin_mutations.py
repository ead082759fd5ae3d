"""CPU: what can the oracle's only reference-held pin see?  (table: DESIGN.md 2, profiles/r3_pin_mutations.json,
made by scripts/pin_mutations.py.)

The reference holds no vectors for this path, only two published renders; tests/test_oracle_golden.py holds the oracle to their
block statistics.  Here the same comparison is run on MUTATED oracles (oracle/liboracle_mut*.so: every quirk of SURVEY 8.1 the
integrator depends on can be flipped to what a clean implementation would do): a quirk the pictures can see must make the pin
test FAIL when flipped -- otherwise the pin does not constrain it and parity on it rests on reading the source alone.  The
blind spots are asserted too, so that nobody reads the pin for more than it is: with EVERY multiply-add contracted
(-ffp-contract=fast, what nvcc's -fmad=true does) the pictures are matched as well as with the one fitted contraction."""
import numpy as np
import pytest

from cuda_path_tracer_amd import generators as G
from cuda_path_tracer_amd.scene import make_config
from oracle import binding as B
from tests.readme_util import CONFIGS, compare, cube_mask_of

SIZE, SPP = 256, 200


def _passes(r):  # the thresholds of test_oracle_against_the_reference_readme_renders
    return r["corr"] > 0.97 and r["mean_dev"] < 0.01 and r["walls"]["sum_dev"] < 0.015 and r["cubes"]["sum_dev"] < 0.03


def _render(cube_obj, key, mutations, variant="mut"):
    desc = G.scene_default(cube_obj)
    B.set_mutations(mutations, variant)
    try:
        o = desc.apply(B.OracleScene(variant=variant))
        o.commit()
        img, _ = o.render(make_config(SIZE, SIZE, spp=SPP, **CONFIGS[key]))
        img = np.nan_to_num(img, nan=0.0, posinf=0.0, neginf=0.0)
        r = compare(key, img, SIZE, SIZE, cube_mask_of(o, desc.camera, SIZE))
        o.close()
    finally:
        B.set_mutations((), variant)
    return r


@pytest.mark.parametrize("key,switch,min_mean_dev", [
    ("simple_d6_200spp", "q5_minx_eps", 0.03),        # the min-X asymmetry of isVertexInBox (mesh.cuh:132): measured 4.8 %
    ("simple_d6_200spp", "dev7_unfused", 0.015),      # the point on a ray with two roundings, as the source text says: 2.4 %
    ("simple_d6_200spp", "q16_roulette_pdf", 0.15),   # roulette probability folded into the pdf (kernel.cu:287-296): 26 %
    ("bidir_3_3_200spp", "q15_single_cos", 0.05),     # cosCam twice in the connection weight (kernel.cu:276): 7.7 %
    ("bidir_3_3_200spp", "q12_uniform_emission", 0.025),  # non-uniform emission with a uniform pdf (light.cuh:35-38): 3.8 %
])
def test_flipping_a_visible_quirk_fails_the_pin(cube_obj, key, switch, min_mean_dev):
    r = _render(cube_obj, key, (switch,))
    print(key, switch, r)
    assert not _passes(r) and r["mean_dev"] > min_mean_dev, r


def test_the_pin_cannot_tell_one_contraction_from_all(cube_obj):
    """Blind spot, asserted: the oracle fuses ONE expression (the point on a ray, deviation 7), chosen because it fits these
    pictures.  A build with every multiply-add contracted fits them just as well -- which of the reference's other products
    and sums nvcc fused is invisible to its published renders, so hit-level parity with the reference stays unpinned."""
    r = _render(cube_obj, "simple_d6_200spp", (), variant="mut_fma")
    print(r)
    assert _passes(r) and r["mean_dev"] < 0.005, r


def test_the_checker_itself_has_no_switches():
    assert B.load().orc_set_mutations(0) == 0
    assert B.load().orc_set_mutations(1) < 0  # liboracle.so is compiled without them: MUT() is a constant false
    with pytest.raises(ValueError):
        B.set_mutations(("no_such_quirk",))
