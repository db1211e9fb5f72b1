"""Flow-matching Euler sampler used by the Wan2.1 pipeline.

The reference never touches the sampler: ``WanVideoGenerator.generate`` forwards nine kwargs to
``WanVideoPipeline.__call__`` [R infinicube/videogen/inference.py:216-226] and the fork's
defaults apply.  Those defaults are the upstream DiffSynth ones restated in SURVEY.md
Appendix A.5/A.6 ([EXT]): ``FlowMatchScheduler(shift=5, sigma_min=0, extra_one_step=True)``,
50 steps, cfg_scale 5.
"""

from __future__ import annotations

from typing import List


def flow_match_sigmas(num_steps: int, shift: float = 5.0, sigma_max: float = 1.0,
                      sigma_min: float = 0.0, denoising_strength: float = 1.0) -> List[float]:
    """sigma_i for i in [0, num_steps): linspace(start, min, N+1)[:-1] then the shift warp."""
    start = sigma_min + (sigma_max - sigma_min) * denoising_strength
    out = []
    for i in range(num_steps):
        s = start + (sigma_min - start) * (i / num_steps)
        out.append(shift * s / (1.0 + (shift - 1.0) * s))
    return out


def round_through_bf16(x: float) -> float:
    """The float nearest-even-rounded to bfloat16 (8 significant bits): 937.5 -> 936.0."""
    import struct
    (u,) = struct.unpack("<I", struct.pack("<f", x))
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return struct.unpack("<f", struct.pack("<I", u))[0]


class FlowMatchScheduler:
    """``reference_rounding``: the timestep fed to the DiT is 1000 sigma rounded to bfloat16, as a pipeline that casts
    ``timestep.to(torch_dtype)`` before the sinusoidal embedding does ([EXT], ORACLE_RISKS.md R1); the sigmas of the
    Euler update stay exact either way."""

    def __init__(self, num_inference_steps: int = 50, shift: float = 5.0, reference_rounding: bool = False):
        self.reference_rounding = reference_rounding
        self.set_timesteps(num_inference_steps, shift)

    def set_timesteps(self, num_inference_steps: int, shift: float = 5.0):
        self.sigmas = flow_match_sigmas(num_inference_steps, shift)
        self.timesteps = [s * 1000.0 for s in self.sigmas]
        if self.reference_rounding:
            self.timesteps = [round_through_bf16(t) for t in self.timesteps]

    def dsigma(self, i: int) -> float:
        """sigma_{i+1} - sigma_i with sigma_N = 0 (the Euler step multiplier)."""
        nxt = self.sigmas[i + 1] if i + 1 < len(self.sigmas) else 0.0
        return nxt - self.sigmas[i]
