"""Model configurations of the benchmark workloads (the reference's shipped YAMLs), matrix peaks and the analytic flop count."""
CONFIGS = {
    "375M": dict(
        in_channels=84, out_channels=84, num_attention_heads=12, attention_head_dim=128, num_layers=2, num_single_layers=4,
        num_refiner_layers=1, mlp_ratio=4, patch_size=1, patch_size_t=1, qk_norm="rms_norm", rope_theta=256.0, rope_axes_dim=(16, 56, 56),
        rope_spatial_grid_start_pos=(-499.5, 5.25), rope_spatial_grid_end_pos=(508.5, 353.25), spatial_deg2rad=True,
        conditioning_tensor_in_channels=84, conditioning_tensor_rope_axes_dim=(16, 56, 56), incl_time_elapsed=True,
    ),  # configs/ladcast_375M.yaml:1-30
}
CONFIGS["1.6B"] = dict(CONFIGS["375M"], num_attention_heads=16, num_layers=5, num_single_layers=10, num_refiner_layers=3)
CONFIG_DCAE_84 = dict(  # configs/DC_AE_84_pretrain.yaml:1-48
    in_channels=89, out_channels=89, latent_channels=84, attention_head_dim=32,
    encoder_block_types=("ResBlock", "ResBlock", "EfficientViTBlock", "EfficientViTBlock"),
    decoder_block_types=("ResBlock", "ResBlock", "EfficientViTBlock", "EfficientViTBlock"),
    encoder_block_out_channels=(252, 504, 504, 1008), decoder_block_out_channels=(252, 504, 504, 1008),
    encoder_layers_per_block=(4, 4, 4, 4), decoder_layers_per_block=(4, 4, 4, 4),
    encoder_qkv_multiscales=((), (), (5,), (5,)), decoder_qkv_multiscales=((), (), (5,), (5,)),
    upsample_block_type="pixel_shuffle", downsample_block_type="pixel_unshuffle", static_channels=5,
)


PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: dense fp32-input MFMA peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense bf16 MFMA peak (the 2:1-sparsity figure is never used)


def model_flops_per_forward(cfg, R, T_in=1, hw=450, split=False):
    """Analytic 2*MAC count of one forward per member; returns (gemm_flops, attn_flops), or with `split` the pair
    ((gemm, attn) of the sample-dependent part, (gemm, attn) of the conditioning path: context embed + token refiner - the part a
    sampler chunk evaluates once per noise level instead of once per network evaluation, LaDCastTransformer3DModel.prepare_conditioning)."""
    D = cfg["num_attention_heads"] * cfg["attention_head_dim"]
    Nx, Nc = R * hw, T_in * hw
    S = Nx + Nc
    F = int(D * cfg["mlp_ratio"])
    lin = lambda m, n, k: 2.0 * m * n * k  # noqa: E731
    cg = lin(Nc, D, 84) + lin(Nc, D, D)  # context embed + refiner proj_in
    cg += cfg["num_refiner_layers"] * (lin(Nc, 3 * D, D) + lin(Nc, F, D) + lin(Nc, D, F))
    ca = cfg["num_refiner_layers"] * 4.0 * Nc * Nc * D
    g = lin(Nx, D, 84)  # sample embed
    g += cfg["num_layers"] * (lin(S, 3 * D, D) + lin(S, D, D) + lin(S, F, D) + lin(S, D, F))
    g += cfg["num_single_layers"] * (lin(S, 3 * D, D) + lin(S, F, D) + lin(S, D, D + F))
    g += lin(Nx, 84, D)
    a = (cfg["num_layers"] + cfg["num_single_layers"]) * 4.0 * S * S * D
    if split:
        return (g, a), (cg, ca)
    return g + cg, a + ca

