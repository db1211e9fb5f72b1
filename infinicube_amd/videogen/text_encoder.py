"""UMT5-XXL text encoder loader (outside the hot loop; stock PyTorch-ROCm).

[R infinicube/videogen/inference.py:68,78] names the file ``models_t5_umt5-xxl-enc-bf16.pth``.
The encoder itself is SURVEY.md §8f row 4 ("next"): not restated in round 1.  ``from_pretrained``
therefore fails loudly instead of substituting anything."""

import glob


def load_umt5_encoder(pattern, device, torch_dtype, tokenizer_config=None):
    files = sorted(glob.glob(pattern))
    if not files:
        raise FileNotFoundError(f"UMT5 encoder checkpoint not found: {pattern!r} (skip_download=True: nothing is fetched)")
    raise NotImplementedError(
        "UMT5-XXL encoder on stock PyTorch-ROCm is a 'next' row (SURVEY.md §8f-4) and not built yet; "
        "construct WanVideoPipeline(text_encoder=...) with any object exposing encode(prompt)->[512,4096]")
