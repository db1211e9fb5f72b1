"""Pipeline factory shared by the rank-0 test process and the worker processes of tests/test_multigpu_workers.py
(resolved in the workers through ICV_WORKER_FACTORY="mgpu_factory:factory"): the product's host code with the TEST-ONLY
oracle operator set and the stand-in encoders, on CPU."""
import torch

from infinicube_amd.videogen import synthetic as syn
from infinicube_amd.videogen.config import TokenGrid, preset
from infinicube_amd.videogen.pipeline import DiTHolder, WanVideoPipeline
from standins import HashTextEncoder, PoolVAE
from oracle_ops import OracleOps

CFG, GRID = preset("tiny"), TokenGrid(9, 64, 96)


def factory(torch_dtype, device, model_configs):
    torch.set_num_threads(2)
    pipe = WanVideoPipeline(device, torch_dtype, DiTHolder(syn.make_dit_state_dict(CFG), CFG), HashTextEncoder(CFG), PoolVAE(),
                            ops=OracleOps())
    pipe.num_inference_steps = 2
    return pipe


def failing_factory(torch_dtype, device, model_configs):
    """Ranks of a worker pool (ICV_WORKER_RANK set) fail while building their pipeline; a plain process builds normally."""
    import os
    if os.environ.get("ICV_WORKER_RANK") is not None:
        raise RuntimeError("synthetic worker failure while loading the checkpoint")
    return factory(torch_dtype, device, model_configs)


def gpu_factory(torch_dtype, device, model_configs):
    """The same tiny pipeline with the PRODUCT operator set on this rank's GPU (tests/test_multigpu_rccl.py: real RCCL ranks)."""
    import os
    from infinicube_amd.videogen.ops import HipOps
    # ICV_TEST_SHARE_GPU=1: every rank on cuda:0 (a 1-GPU box, gloo backend) instead of cuda:LOCAL_RANK
    dev = "cuda:0" if os.environ.get("ICV_TEST_SHARE_GPU") == "1" else WanVideoPipeline.resolve_device(device)
    pipe = WanVideoPipeline(dev, torch_dtype, DiTHolder(syn.make_dit_state_dict(CFG), CFG), HashTextEncoder(CFG), PoolVAE(), ops=HipOps(dev))
    pipe.num_inference_steps = 2
    return pipe


def small_wan_vae():
    """The real WanVAE code (tiled, channels as in the public architecture but 8 base channels) with seeded random weights, fp32 on CPU."""
    from infinicube_amd.videogen.vae import WanVAE, WanVAENet
    torch.manual_seed(11)
    return WanVAE(WanVAENet(dim=8), "cpu", torch.float32)


def real_vae_factory(torch_dtype, device, model_configs):
    """`factory` with the product's tiled Wan-VAE instead of the stand-in: in a multi-rank run its tiles are dealt to the ranks."""
    torch.set_num_threads(2)
    pipe = WanVideoPipeline(device, torch_dtype, DiTHolder(syn.make_dit_state_dict(CFG), CFG), HashTextEncoder(CFG), small_wan_vae(),
                            ops=OracleOps())
    pipe.num_inference_steps = 2
    return pipe


def gpu_real_vae_factory(torch_dtype, device, model_configs):
    """`gpu_factory` with the PRODUCT configuration of the tiled Wan-VAE (bf16, NDHWC, folded pad, HIP norm kernel; 32 base channels,
    seeded weights) instead of the pooling stand-in: in a multi-rank run its tiles are dealt to the ranks on the GPU."""
    import os
    from infinicube_amd.videogen.ops import HipOps
    from infinicube_amd.videogen.vae import WanVAE, WanVAENet
    dev = "cuda:0" if os.environ.get("ICV_TEST_SHARE_GPU") == "1" else WanVideoPipeline.resolve_device(device)
    torch.manual_seed(11)
    vae = WanVAE(WanVAENet(dim=32), dev, torch.bfloat16)
    pipe = WanVideoPipeline(dev, torch_dtype, DiTHolder(syn.make_dit_state_dict(CFG), CFG), HashTextEncoder(CFG), vae, ops=HipOps(dev))
    pipe.num_inference_steps = 2
    return pipe


class _FlakyPipeline(WanVideoPipeline):
    """Rank 1's SECOND request fails in the middle of the pipeline call (after the ranks have entered their collectives)."""
    calls = 0

    def __call__(self, *a, **k):
        import os
        type(self).calls += 1
        if os.environ.get("ICV_WORKER_RANK") == "1" and type(self).calls == 2:
            raise RuntimeError("synthetic failure in the middle of request 2 on rank 1")
        return super().__call__(*a, **k)


def flaky_factory(torch_dtype, device, model_configs):
    torch.set_num_threads(2)
    pipe = _FlakyPipeline(device, torch_dtype, DiTHolder(syn.make_dit_state_dict(CFG), CFG), HashTextEncoder(CFG), PoolVAE(), ops=OracleOps())
    pipe.num_inference_steps = 2
    return pipe
