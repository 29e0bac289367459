"""Schedule / materialisation knobs of the host side, in ONE place.  The defaults are the measured best (DESIGN.md section 2
'what was tried'); TrainStep(options={...}) overrides them per instance.  Environment overrides (RD_*) are honoured only in
debug mode (RAMDSIR_DEBUG_LIB=1, the library build whose kernels' dispatch switches are live too): a production run reads
no tuning environment."""
import os

DEFAULTS = dict(
    mat_min_c=0,          # >0: store relu(bn(z)) once for layers with at least this many channels (measured a net loss: off)
    mat_dz_min_c=0,       # >0: store the BN-backward gradient dz once (rd_bn_apply) for layers with at least this many channels.  Off since
                          # round 4: conv_ws_kernel<2,TS,true> computes dz from (z, g) in its loader warps, so the gradient launch and the
                          # weight gradient both read the two operands and the extra pass (14 launches, 0.25 ms of main-lane time) is gone:
                          # 4.36 -> 4.23 ms/step together with side_cus 128 -> 96 (docs/experiments.md, round-4 rows)
    mat_dz_wide=True,     # with mat_dz_min_c > 0: ... and for a 32-channel layer whose gradient launch is 64-wide (engine.Plan.build)
    up_mat=True,          # store y = up2(t) once (rd_up_stats) instead of interpolating t in the loaders of the 3x3 conv behind it (round 5: 4.19 vs
                          # 5.0+ ms/step, docs/experiments.md)
    pool_mat=True,        # store the 2x2 max-pool in front of ConvD levels 2-5 once (rd_pool_fwd / rd_pool_bwd)
    fused_bwd=True,       # <= 32-channel 3x3 convs (bf16): dgrad + weight gradient in one launch (csrc/conv_fused.hip)
    split_wide_dgrad=False,  # a one-chunk gradient launch with 33..64 output channels as two launches of the small-channel kernel
                          # over row blocks of the packed weights (rd_conv_t.w_tap_rows).  A gain (-0.05 ms) while the 64-wide
                          # conv_pf_kernel's gradient epilogue cost 33 000 cycles per workgroup; since that was fixed (DESIGN.md
                          # section 7) the ONE 64-wide launch wins: dec.convu1.conv1 dgrad 103 us against 57 + 54, and it reads dz
                          # once -- step 4.61 -> 4.57 ms (debug library, alternating runs)
    side_streams=1,       # HIP streams for the weight-gradient launches
    rec_cus=-1,           # compute units the restoration-decoder lane's persistent launches may take (0: all, -1: half of
                          # the device = 128 on MI355X, where the numbers below were measured); the lane ends
                          # 1.4 ms before the main one, so it can run narrower: 5.59 -> 5.55 ms/step (96: 5.57, 64: 5.70)
    side_cus=96,          # compute units a weight-gradient launch may take while it runs beside the dgrad chain (fork=True;
                          # 0: all, -1: half of the device); its persistent workgroups cannot share a CU with the chain's kernels.
                          # Round-4 sweep with the gradient chain on conv_ws_kernel<2> at dgrad_cus=160 (160 + 96 = the device):
                          # 64: 4.29, 80: 4.23, 96: 4.23 / 4.20, 112: 4.36, 128: 4.36 ms/step; the HBM-bound 16-channel launches keep the whole GPU
    store_wgrad_operands=0,      # bit 0 / bit 1: launches on conv_ws_kernel also store what their loaders stage (act(bn(z)) of the forward launch / dz of
                          # the gradient launch: rd_src_t.out) and the layer's weight gradient copies stored operands instead of transforming.
                          # MEASURED A LOSS in the step and therefore off: the >= 64-channel weight gradients get 25-30 % faster (lane 1.91 ->
                          # 1.62 ms alone) but the stores cost the main lane's launches as much HBM time (64-wide convs 1.22 -> 1.39 ms alone):
                          # 4.283 (>= 128 channels) / 4.358 (all) / 4.291 (forward operand only) against 4.269 ms/step without
    store_wgrad_min_c=128,       # ... for operands of at least this many channels
    fork=True,            # eager launch over main / side / rec streams (False: one stream)
    rec_lane=True,        # the restoration decoder branch on its own stream
    rec_wgrad_late=True,  # its stand-alone weight gradients behind the join with the main lane instead of inside its dgrad chain
    graph_fork=False,     # capture(): keep the forks as graph branches (slower on ROCm 7: DESIGN.md section 3)
    ddp_own_comm_stream=-1,     # data parallel: where the all-reduces are launched from.  0: the weight-gradient lane (RCCL runs them on
                          # its own stream anyway); 1: a stream of their own (a fifth stream: it can alias a lane's hardware queue on a
                          # 4-queue stack, ddp.py); -1: MEASURED at start-up like the lanes are (DataParallelStep.pick_comm: a few
                          # steps each way on this rank's real process group, the slower rank decides) -- no multi-GPU node was
                          # available to fix the choice at build time
    dgrad_cus=160,        # compute-unit budget of the seg plan's >= 64-channel gradient launches (they run on the persistent whole-CU
                          # kernel conv_ws_kernel<2> since round 4): leaves 96 CUs to the weight-gradient / restoration lanes beside
                          # them.  0 (all): 4.48 ms/step, 224: 4.46, 192: 4.44, 160: 4.43, 128: 4.51 (scripts/attic/sweep_ws2.sh)
    launch_threads=-1,    # (-1: measured at start-up by the data-parallel wrapper, DataParallelStep.pick_launch_threads; a single process keeps one
                          # thread) rd_run_list_threads: the side / rec lanes' launches are enqueued by worker threads of the library, in parallel
                          # with the main lane's (host enqueue 0.85 -> ~0.4 ms per step; the GPU executes the same graph)
    fold_finalize=6,      # BatchNorm finalize launches folded into the prologue of the launch that first reads the coefficients (rd_src_t.fin,
                          # csrc/bn_fin.h; statistic sums over 8 slot copies instead of 64).  0: explicit launches; 1: all 76 of the step;
                          # 3: the forward ones; 4: the backward ones; 6 (default): the forward ones + the backward ones of layers whose
                          # gradient and weight gradient are ONE launch; 5: 6 + rd_up_bwd; 2: timing experiment, no finalize at all.
                          # Same box, alternating (scripts/options_ab.py): explicit 4.267, 6: 4.147 ms/step; 1: 4.230 against 4.213 and
                          # 3: 4.161 -- a stand-alone weight gradient beside its gradient launch repeats the prologue on its own lane,
                          # and the many-workgroup rd_up_bwd pays it per workgroup; without ANY finalize the step takes 4.03.  Also measured:
                          # the explicit backward launch moved to the weight-gradient lane with the gradient launch deriving P / Q / R itself
                          # (4.542 against 4.423 for 6 and 4.531 explicit, a slower box)
    fold_fwd_kinds=7,     # which forward consumers take a folded finalize: 1 small-channel convs, 2 wide convs, 4 the max-pool
    fold_wgrad_behind=False,   # with fold_finalize: a stand-alone weight gradient is enqueued BEHIND its layer's gradient launch (which owns the
                          # folded BatchNorm-backward finalize) instead of in front of it with a prologue of its own
    conv_nb1_below=300,   # mirrors csrc/conv_big.hip: 64-wide launches below this many workgroups run 32-wide tiles (meta only)
)
_ENV = dict(split_wide_dgrad='RD_SPLIT_WIDE_DGRAD', mat_min_c='RD_MAT_MINC', mat_dz_min_c='RD_MAT_DZ_MINC', mat_dz_wide='RD_MAT_DZ_WIDE', pool_mat='RD_POOL_MAT', up_mat='RD_UP_MAT', side_streams='RD_SIDE_STREAMS', side_cus='RD_SIDE_CUS', rec_cus='RD_REC_CUS',
            ddp_own_comm_stream='RD_DDP_OWN_COMM', fused_bwd='RD_FUSED_BWD_HOST', fork='RD_FORK', store_wgrad_operands='RD_STORE_WGRAD_OPS', launch_threads='RD_LAUNCH_THREADS', dgrad_cus='RD_DGRAD_CUS', rec_lane='RD_REC_LANE', rec_wgrad_late='RD_REC_WGRAD_LATE', graph_fork='RD_GRAPH_FORK', conv_nb1_below='RD_CONV_NB1_BELOW', fold_finalize='RD_FOLD_FINALIZE', fold_wgrad_behind='RD_FOLD_WGRAD_BEHIND', fold_fwd_kinds='RD_FOLD_FWD_KINDS')


def options(over=None):
    o = dict(DEFAULTS)
    if os.environ.get('RAMDSIR_DEBUG_LIB') == '1':
        for k, name in _ENV.items():
            if name in os.environ:
                o[k] = type(DEFAULTS[k])(int(os.environ[name]))
    if over:
        over = {k: v for k, v in over.items() if v is not None}
        unknown = set(over) - set(DEFAULTS)
        if unknown:
            raise KeyError('unknown tuning option(s): %s' % sorted(unknown))
        o.update(over)
    return o


def cu_budget(value, device):
    """side_cus / rec_cus / dgrad_cus -> compute units of THIS device.  The positive defaults are the measured optima on the 256-CU
    MI355X and are scaled with the device's compute-unit count (hipDeviceProp multiProcessorCount; a multiple of 8 = whole XCD rows),
    -1 = half of the device, 0 = no budget.  device None: the value as given (host-side tests)."""
    if value == 0 or device is None:
        return int(value)
    import torch
    n = torch.cuda.get_device_properties(device).multi_processor_count
    if value < 0:
        return n // 2
    return int(value) if n == 256 else max(8, int(round(value * n / 256.0 / 8.0)) * 8)
