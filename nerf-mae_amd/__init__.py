"""nerf_mae_amd -- MI355X-native (gfx950) engine for the NeRF-MAE 3-D Swin MAE pre-training hot path.

Public surface mirrors the reference (nerf_mae/model/mae/swin_mae3d.py, run_swin_mae3d.py):
`SwinTransformer_MAE3D` / `SwinTransformer_MAE3D_New`, same constructor kwargs, `forward(list_of_grids, is_eval)`
contract and `state_dict()` keys.  All arithmetic runs in libnerfmae_hip.so (hand-written HIP); there is no CPU
fallback and importing the model without the built library fails loudly."""
__version__ = "0.1.0"
