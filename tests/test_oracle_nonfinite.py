"""CPU: the two ways the reference's arithmetic makes a sample non-finite, on the oracle alone (the GPU side of the
same (seed, pixel, sample) triples is tests/test_gpu_full_configs.py).  See that file's header for the expressions."""
import numpy as np

from cuda_path_tracer_amd import generators as G
from cuda_path_tracer_amd.scene import make_config
from oracle.binding import OracleScene


def test_zero_light_pdf_makes_the_bidirectional_sample_infinite(cube_obj):
    o = G.scene_default(cube_obj).apply(OracleScene())
    o.commit()
    for smp, pix in ((109, 688568), (113, 398430)):
        cfg = make_config(1024, 1024, spp=200, cam_depth=3, light_depth=3, sample_begin=smp, sample_end=smp + 1)
        rgb, probe = o.explain_sample(cfg, pix, smp)
        assert not np.isfinite(rgb).any()
        # kernel.cu:277  light_throughput[j] / light_pdf[j]  with light_pdf[j] == 0 (SampleDiffuse returned z/pi == 0)
        assert probe["first_cause"] == 2 and probe["light_vertex"] in (1, 2) and probe["operand"][1] == 0.0
        assert probe["zero_pdf_vertices"] == 2 and probe["sqrt_negative"] == 0
        # a neighbouring pixel of the same sample index is an ordinary finite sample
        rgb2, probe2 = o.explain_sample(cfg, pix + 1, smp)
        assert np.isfinite(rgb2).all() and probe2["first_cause"] == 0


def test_sqrt_of_a_negative_in_sample_diffuse_reaches_the_image_only_through_the_environment():
    # C4's scene and frame (3840x2160, simple depth 6, cube map): samples found by scripts/find_nonfinite.py
    o = G.scene_env_sphere(8, env_f32=G.synthetic_env_faces(64)).apply(OracleScene())
    o.commit()
    for smp, pix in ((6, 4023932), (10, 2399386)):
        for env in (True, False):
            cfg = make_config(3840, 2160, spp=1024, cam_depth=6, use_envmap=env, sample_begin=smp, sample_end=smp + 1)
            rgb, probe = o.explain_sample(cfg, pix, smp)
            # samplers.cuh:19  z = sqrtf(1 - x*x - y*y) of a negative; no contribution term is evaluated with a NaN operand:
            # the NaN arrives through (cam_throughput / cam_pdf) * envMap(dir) on the miss of the NaN ray
            assert probe["sqrt_negative"] == 1 and probe["first_cause"] == 0
            assert np.isnan(rgb).all() if env else np.isfinite(rgb).all()
