"""CPU oracle for the latent-diffusion denoising hot path.

TEST INFRASTRUCTURE ONLY.  This package restates, in plain PyTorch CPU ops, the
algorithm of the reference's sampling path (U-Net noise predictor, DDPM/DDIM
schedulers, K-samplers).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the shipped package
``diffusion_model_amd`` never does.

Pinning: every function here is checked against the reference imported from
``/root/reference`` by ``tools/gen_golden.py`` (run in the build container);
the resulting vectors live under ``tests/golden/`` and
``tests/test_oracle_golden.py`` re-checks the oracle against them wherever the
suite runs.
"""
