"""Encoder factory (reference: tools/encoding.py:52-123).  FreqEncoder stays plain torch -- it is
only used by the `mlp` model type and is not on the HIP path."""
import torch
import torch.nn as nn


class FreqEncoder(nn.Module):
    """NeRF positional encoding (reference: tools/encoding.py:6-49): [x, sin(f x), cos(f x), ...] with
    f = 2^linspace(0, max_freq_log2, N_freqs)."""

    def __init__(self, input_dim, max_freq_log2, N_freqs, log_sampling=True, include_input=True):
        super().__init__()
        self.input_dim = input_dim
        self.include_input = include_input
        self.output_dim = (input_dim if include_input else 0) + input_dim * N_freqs * 2
        if log_sampling:
            bands = 2.0 ** torch.linspace(0.0, max_freq_log2, N_freqs)
        else:
            bands = torch.linspace(2.0 ** 0.0, 2.0 ** max_freq_log2, N_freqs)
        self.freq_bands = bands.numpy().tolist()

    hip_encode = None  # set by get_encoder when the operator set has a fused kernel (pvd_freq_encode)

    def forward(self, x, **kwargs):
        if self.hip_encode is not None and x.is_cuda and not x.requires_grad and x.dtype == torch.float32 and len(self.freq_bands) <= 16:
            # one launch instead of 4 per frequency + a concatenation
            return self.hip_encode(x.reshape(-1, self.input_dim).contiguous(), self.freq_bands, self.include_input).view(*x.shape[:-1], self.output_dim)
        parts = [x] if self.include_input else []
        for f in self.freq_bands:
            parts.append(torch.sin(x * f))
            parts.append(torch.cos(x * f))
        return torch.cat(parts, dim=-1)


def get_encoder(ops, encoding, input_dim=3, multires=6, degree=4, num_levels=14, level_dim=2, base_resolution=16,
                log2_hashmap_size=19, desired_resolution=4096, align_corners=False, **kwargs):
    if encoding == "None":
        return (lambda x, **kw: x), input_dim
    if encoding == "frequency":
        enc = FreqEncoder(input_dim=input_dim, max_freq_log2=multires - 1, N_freqs=multires, log_sampling=True)
        enc.hip_encode = getattr(ops, "freq_encode", None)
    elif encoding == "sphere_harmonics":
        enc = ops.SHEncoder(input_dim=input_dim, degree=degree)
    elif encoding in ("hashgrid", "tiledgrid"):
        enc = ops.GridEncoder(input_dim=input_dim, num_levels=num_levels, level_dim=level_dim, base_resolution=base_resolution,
                              log2_hashmap_size=log2_hashmap_size, desired_resolution=desired_resolution,
                              gridtype="hash" if encoding == "hashgrid" else "tiled", align_corners=align_corners)
    else:
        raise NotImplementedError(encoding)
    return enc, enc.output_dim
