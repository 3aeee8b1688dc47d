"""rfx_amd — Python host side of the MI355X-native SSGI hot path (tests / bench plumbing).

    abi      ctypes mirror of include/rfx.h + loader of csrc/librfx_hip.so (no fallback)
    context  Context: one rfx_ctx (a GPU, or one row tile of the frame)
    effect   the reference's operator surface: SSGIEffect, TRAAEffect, Denoiser, the passes
    tiling   row tiles over the GPUs of a node, halo exchange through torch.distributed
    scene    synthetic G-buffer dumps in the reference's texel formats
    dump     on-disk dump format shared with the Node host (../js)
"""
__all__ = ["abi", "context", "effect", "tiling", "scene", "dump"]
