"""Every golden fixture stores the SHA-256 of the synthetic inputs and weights the reference's modules were run on
(tests/golden/make_golden*.py).  The generators are bit-reproducible by construction (CPU torch.Generator draws, integer /
element-wise float64 image synthesis — gen6d_amd/synth.py), so the same hashes must come out on any host: here, and on the GPU box
where the `-m gpu` tests assert them again next to every `vs reference golden` comparison."""
import numpy as np
import pytest

from conftest import assert_pinned
from gen6d_amd import synth

CASES = {
    "det_small": lambda g: (synth.detector_case(8, 128, 128), synth.synth_state_dict("detector")),
    "det_mid": lambda g: (synth.detector_case(32, 160, 192), synth.synth_state_dict("detector")),
    "det_head": lambda g: (synth.detector_case(32, 480, 640), synth.synth_state_dict("detector")),
    "sel_small": lambda g: (synth.selector_case(8, 5), synth.synth_state_dict("selector", an=5)),
    "sel_mid": lambda g: (synth.selector_case(16, 5), synth.synth_state_dict("selector", an=5)),
    "sel_head": lambda g: (synth.selector_case(64, 5), synth.synth_state_dict("selector", an=5)),
    "sel_128x5": lambda g: (synth.selector_case(128, 5), synth.synth_state_dict("selector", an=5)),
    "sel_32x5": lambda g: (synth.selector_case(32, 5), synth.synth_state_dict("selector", an=5)),
    "sel_64x36": lambda g: (synth.selector_case(64, 36), synth.synth_state_dict("selector", an=36)),
    "ref_step": lambda g: (synth.refiner_case(), synth.synth_state_dict("refiner")),
    "ref_grids": lambda g: (synth.refiner_case(), synth.synth_state_dict("refiner")),
}


@pytest.mark.parametrize("tag", sorted(CASES))
def test_fixture_inputs_are_pinned(golden, tag):
    g = golden(tag)
    inputs, weights = CASES[tag](g)
    assert_pinned(g, inputs, weights, tag)


def test_rotated_copies_are_integer_exact():
    """rotated_copies: rotation by 0 is the identity, by +-90 degrees an exact transpose/flip (no interpolation), and the output
    does not depend on float rounding of the blend (integer arithmetic on 1/256-pixel weights)."""
    imgs = synth.synth_images(3, 128, 128, 9)
    r = synth.rotated_copies(imgs, 5)
    assert r.dtype == np.uint8 and r.shape == (5, 3, 128, 128, 3)
    assert np.array_equal(r[2], imgs)
    assert np.array_equal(r[0], np.rot90(imgs, k=1, axes=(1, 2))) or np.array_equal(r[0], np.rot90(imgs, k=-1, axes=(1, 2)))
    assert np.array_equal(r[4], np.rot90(r[0], k=2, axes=(1, 2)))
