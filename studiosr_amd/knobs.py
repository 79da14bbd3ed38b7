"""Every environment switch the package reads, in ONE place (tests/test_host_cpu.py holds the sources to this table: a switch that is not listed fails the
CPU suite, so the set cannot grow silently).  Kinds:
  select  an alternative implementation of the same result, kept because a parity test compares the two forms or because it is the fallback of a geometry
  tune    a size threshold / count measured on MI355X (the default is the measured optimum; the variable exists for re-tuning on other parts)
  diag    diagnostics, never needed for results
Round 5 folded thirteen switches whose experiment had concluded into constants (SR_SWIN_BLOCK, SR_WG_XCD, SR_CONV_XCD, SR_CONV_NARROW, SR_BGEMM_BN, SR_BGEMM_NATURAL,
SR_BGEMM_DIRECT, SR_COLSUM_BLOCKS, SR_TR_QVAR, SR_TAIL_POOL_LDS, SR_MLP_ROWS, SR_QKV_N1, SR_HAT_SIDE_FIRST, SR_WGRAD_STREAM)."""

KNOBS = {
    # ---- library / runtime
    "SR_LIB_PATH": ("", "diag", "load this libstudiosr_hip.so instead of studiosr_amd/lib/ (variant libraries of tools/exp3.sh)"),
    "SR_WS_GUARD": ("", "diag", "workspace: poison + check guard bands around every buffer"),
    "SR_WS_BUDGET_MB": ("16384", "tune", "workspace: refuse to grow beyond this many MiB"),
    # ---- SwinIR (config 3)
    "SR_BLOCK_WGS": ("0", "tune", "sr_swin_block grid: 0 one workgroup per window (default), N at most N persistent workgroups, -2 as many as are resident at once"),
    "SR_SWIN_LIGHT": ("1", "select", "embed-60 geometry: the one-launch sr_swin_light kernel (0: QKV / attention / tail launches; bit-compared in tests)"),
    "SR_SWIN_PARTS": ("0", "tune", "a batch as N part batches on the model's streams inside a graph capture (0: two from 16 tiles on)"),
    "SR_SWIN_QKV": ("1", "select", "stream-form QKV kernel (0: the generic GEMM with LayerNorm prologue)"),
    "SR_SWIN_TAIL": ("1", "select", "stream-form projection + MLP kernel (0: projection GEMM + sr_mlp)"),
    "SR_QKV_FRAG": ("1", "select", "q / k / v^T handed to the attention kernel in fragment order (0: row-major; bit-compared)"),
    "SR_STRIPS_OVERLAP": ("1", "select", "row strips: SW-MSA halo exchange on a side stream beside the interior windows (0: serial order; same bits)"),
    # ---- HAT inference
    "SR_HAB_MID": ("1", "select", "window attention + CAB as ONE launch (sr_hab_mid); also read by the training forward"),
    "SR_ATTN_X3": ("1", "select", "precision fp32x3: window attention on split-operand bf16 MFMAs (0: exact fp32 MFMAs; compared in tests)"),
    "SR_ATTN_LDS": ("1", "select", "window attention with K / V^T / distinct bias tiles staged in LDS (0: flash form; compared in tests)"),
    "SR_ATTN_QKV": ("auto", "tune", "attention workgroups project their own head's q / k / v (auto: up to 128 (window, head) items)"),
    "SR_OCA_LDS": ("1", "select", "overlapping cross attention with K / V^T / table in LDS (0: flash form; compared in tests)"),
    "SR_CAB_X3": ("0", "select", "precision fp32x3: the CAB's two convs as ONE split-operand launch (tested; measured slower than the two sr_conv3x3 launches at HAT's sizes: off)"),
    "SR_CAB_FUSED": ("1", "select", "the CAB's two convs as one launch (0: two sr_conv3x3 launches)"),
    "SR_CAB_ROWS8_FROM": ("16384", "tune", "CAB tiles of 14 x 8 outputs from this many pixels on"),
    "SR_TAIL_GATE": ("1", "select", "channel-attention gate recomputed inside sr_swin_tail (0: sr_channel_gate launch)"),
    "SR_TAIL_QKV": ("1", "select", "sr_swin_tail goes on with the next block's LayerNorm1 + QKV"),
    "SR_TAIL_OCA": ("1", "select", "... and, for the last HAB of a group, with the OCAB's QKV in the zero-bordered layouts"),
    "SR_TAIL_WG32_UPTO": ("256", "tune", "32-token tail workgroups up to this many 64-token workgroups"),
    "SR_TAIL_MT2_BELOW": ("", "tune", "library: force the two-row-tile tail instantiation below this workgroup count"),
    "SR_HAT_DUAL": ("1", "select", "CAB branch of a HAB on a side stream"),
    "SR_HAT_PARTS": ("0", "tune", "HAT batch as N part batches inside a graph capture (0: automatic)"),
    "SR_HAT_PART_MIN": ("4", "tune", "images per HAT part batch"),
    # ---- convs / RCAN
    "SR_CONV_BIG_MIN": ("224", "tune", "library: the wide-tile conv kernel from this many tiles on"),
    "SR_CONV_TH4_BELOW": ("256", "tune", "library: 4-row conv tiles below this many 8-row tiles"),
    "SR_RCAB_X3": ("1", "select", "precision fp32x3: the RCAB's conv pair as ONE split-operand launch (0: two sr_conv3x3 launches; compared in tests)"),
    "SR_RCAN_PARTS": ("0", "tune", "RCAN batch as N part batches (0: automatic inside a graph capture)"),
    # ---- training (config 5)
    "SR_FAST_TRAIN": ("1", "select", "fused HAT training path (0: the generic engine, the exact-fp32 parity path; compared in tests)"),
    "SR_FAST_FULL": ("1", "select", "head and tail of the model on fused launches too (0: only the RHAGs)"),
    "SR_FAST_OCAB": ("1", "select", "the OCAB on fused launches"),
    "SR_FAST_NODES": ("1", "select", "the step as a chain of autograd nodes, one per RHAG (0: one node; compared in the DDP overlap test)"),
    "SR_TR_PLAN": ("1", "select", "launch sequences recorded once and replayed by sr_plan_run (0: enqueued from Python every step)"),
    "SR_TR_WG_SIDE": ("0", "select", "weight-gradient launches on a stream of their own with doubled operand sets (measured +-0)"),
    "SR_TR_CAB_BWD_FUSED": ("1", "select", "the CAB's data gradient (conv, GELU', conv) as one sr_cab_fused launch in its backward form (0: three launches; compared by the fused tests)"),
    "SR_TR_CONV_WG_SIDE": ("1", "select", "the CAB convs' weight-gradient launch on the backward's side stream beside sr_tr_qkv_bwd and the nn.Linear weight gradients"),
    "SR_TR_BWD_DUAL": ("1", "select", "CAB branch of a HAB's backward on a side stream"),
    "SR_TR_ATTN_LDS": ("1", "select", "window-attention backward as one LDS-form launch (0: two register passes; both tested against torch)"),
    "SR_TR_OCA_LSE": ("1", "select", "the OCAB's forward keeps its log-sum-exp and the backward's pass Q runs tile by tile at two workgroups per CU (0: pass Q recomputes the softmax)"),
    "SR_TR_OCA_KV_LDS": ("1", "select", "library: OCAB pass KV with the query side and the bias table in LDS (0: the generic register pass; compared by the fused tests)"),
    "SR_TR_OCA_LDS": ("1", "select", "OCAB pass Q in LDS form"),
    "SR_TR_MIDPRE": ("1", "select", "the CAB's conv1 pre-activation kept by the forward (0: recomputed in the backward)"),
    "SR_TR_GROUPS": ("21", "tune", "window groups of the OCAB's pass Q (default 512 // (6 heads x 4), rounded down to a divisor of the window count; 256 // 24 with SR_TR_OCA_LSE=0)"),
    "SR_WG_WIDE": ("1", "select", "library: nn.Linear weight gradients on 192 x 96 tiles (sr_tr_wgrad_wide_kernel; 0: the 64 x 64 tiles; both tested against torch)"),
    "SR_TR_FINALIZE_LONG": ("64", "tune", "gradient partial sums with at least this many slices are reduced by eight lanes per element (0: one thread per element)"),
    "SR_WG_KS": ("16", "tune", "token slices of the weight-gradient GEMMs"),
    "SR_WG_HALO": ("1", "select", "3x3 weight gradients on 2-D patches with one staged halo for all nine taps"),
    "SR_WG_HALO_STEPS": ("16", "tune", "patches per slice of the halo form"),
}
