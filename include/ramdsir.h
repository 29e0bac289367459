/* ramdsir.h -- C ABI of libramdsir_hip.so: the MI355X (gfx950) hot path of RAM-DSIR training.
 *
 * The reference (zzzqzhou/RAM-DSIR) is pure Python on top of PyTorch; it has no FFI.  The boundary a
 * maintainer binds is therefore the set of ATen/cuDNN operators its nn.Modules dispatch to
 *  plus the numpy RAM trio.  Every entry point below names the reference site(s) it
 * replaces (paths relative to the reference checkout).  INTEGRATION.md shows the ctypes stub that
 * plugs them under code/networks/unet.py and code/dataset/fundus.py.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the name ends in _host; the library never allocates
 *     tensors (the caller owns all memory; workspaces are passed in);
 *   - activations/gradients are NHWC, element type `dtype` (RD_F32 or RD_BF16); parameters, BN
 *     statistics, losses and Adam state are always fp32, OIHW like torch;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*), returns 0 or a
 *     hipError_t value, and is hipGraph-capturable (no allocation, no sync inside);
 *   - a "group" is a set of consecutive images that share BatchNorm statistics: the two encoder /
 *     seg-decoder passes of one step (img, img_freq) are two groups of one launch; the per-domain
 *     slices of the restoration decoder are one group per domain (DomainSpecificBatchNorm2d).
 */
#ifndef RAMDSIR_H
#define RAMDSIR_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define RD_F32 0
#define RD_BF16 1
#define RD_MAX_GROUPS_C 16
/* BatchNorm sums are accumulated with atomics; to keep thousands of workgroups off the same few addresses
 * every statistics buffer has RD_STAT_SLOTS copies, [G][RD_STAT_SLOTS][C][2]; a workgroup adds into slot
 * (its tile index mod RD_STAT_SLOTS) and the finalize kernels sum the slots.
 * Numerics (ATen's batch_norm uses a mean-then-deviation / Welford reduction, never E[x^2]-E[x]^2 in fp32): a workgroup
 * sums its tile in fp32, but (a) the forward sums are those of the conv result WITHOUT its bias (the bias -- the one
 * structural source of |mean| >> sigma -- is added back to the mean by rd_bn_finalize_fwd), and (b) everything across
 * workgroups -- the slot atomics, the slot sum, var = E[x^2]-E[x]^2 and sum g*z - mean * sum g -- is fp64. */
#define RD_STAT_SLOTS 64
/* A launch may be told to spread its sums over the FIRST n of the RD_STAT_SLOTS copies only (rd_conv_t.stat_slots, the stat_slots
 * arguments of rd_up_stats / rd_pool_bwd; 0 = all): the persistent kernels of the training step add once per workgroup, not once per
 * tile, so 8 copies keep them apart -- and a consumer that derives the BatchNorm coefficients itself (rd_src_t.fin below) then reads 8
 * copies per channel instead of 64.  The layout [G][RD_STAT_SLOTS][C][2] does not change; unused copies stay zero, and the explicit
 * finalize launches sum all of them. */
#define RD_STAT_SLOTS_FOLD 8

/* rd_src_t.fin_flags */
enum {
    RD_FIN_OWNER = 1   /* this launch also does what happens once per BatchNorm call: saved mean / invstd, running statistics and
                          num_batches_tracked (forward); dgamma / dbeta (backward) */
};

/* how a conv reads one of its (virtual) input tensors -- the producer's BatchNorm + activation
 * (+ max-pool / bilinear x2) are applied while the tile is staged into LDS ("normalise on read") */
enum {
    RD_SRC_RAW = 0,    /* v = x                                             (image, dlogits)             */
    RD_SRC_AFF = 1,    /* v = x*scale+shift                                 (ConvD.bn1: no activation)   */
    RD_SRC_AFFACT = 2, /* v = act(x*scale+shift)                            (BN+ReLU, unet.py:64-70)     */
    RD_SRC_POOL = 3,   /* v = max2x2(act(x*scale+shift)), stored at 2H x 2W (unet.py:56 MaxPool2d(2))    */
    RD_SRC_UP = 4,     /* v = act(bilerp2x(x)*scale+shift), stored at H/2 x W/2 (unet.py:84-86, 1x1 conv
                          commuted below the upsample: conv1x1(up(a)) == up(conv1x1(a)))                 */
    RD_SRC_BNBWD = 5   /* v = P*g + Q*z + R  (BatchNorm backward folded into the read; ptr=g, ptr2=z)    */
};

typedef struct {
    const void* ptr;     /* NHWC tensor [N][Hs][Ws][C]                                   */
    const void* ptr2;    /* RD_SRC_BNBWD: the raw forward tensor z                        */
    const float* scale;  /* [G][C]  (BNBWD: P)                                            */
    const float* shift;  /* [G][C]  (BNBWD: R)                                            */
    const float* q;      /* [G][C]  BNBWD only                                            */
    void* out;           /* NULL, or a tensor of ptr's shape: the launch also WRITES the values it stages from this source --
                          * act(scale x + shift), or dz = P g + Q z + R -- there (storage dtype), every pixel once.  Honoured by
                          * the launches for which rd_conv_honours_src_out() returns 1, ignored by all others.  The weight
                          * gradient of the same layer then reads stored operands (RD_SRC_RAW) instead of repeating the
                          * transform in its loader, which is what bounds it                          */
    int32_t mode;
    int32_t C;
    float slope;         /* 0 = ReLU, 0.01 = LeakyReLU (unet.py:47-50)                    */
    int32_t n_off;       /* image offset into ptr (rec decoder reads images 8.. of x5)    */
    int32_t g_fixed;     /* >=0: use this group row of scale/shift instead of the image's */
    int32_t fin_flags;   /* RD_FIN_* when fin != NULL */
    const void* fin;     /* NULL, or the rd_bn_fwd_t (modes AFF / AFFACT / POOL / UP: it produces scale / shift) resp. rd_bn_bwd_t
                          * (RD_SRC_BNBWD: P / Q / R) that describes the BatchNorm finalize of this source (a HOST struct like this
                          * one, read when the entry point is called): the launch runs it as its own prologue -- every workgroup derives
                          * the coefficient vectors from the statistic slots and writes them where the descriptor says, which is
                          * where scale / shift / q of this source must point -- instead of an rd_bn_finalize_* launch in front of it
                          * (training mode, BatchNorm only, at most 8 groups, one such source per launch).  Launches that read the same
                          * coefficients LATER in stream order need no fin; launches that may run CONCURRENTLY (another stream) each
                          * carry the fin, exactly one of them with RD_FIN_OWNER.  Honoured by rd_conv, rd_wgrad (dz only) and
                          * rd_conv_bwd_fused (the gradient descriptor's source); -3 for a fin the launch cannot take */
} rd_src_t;

/* where a dgrad launch puts the gradient w.r.t. its (virtual) input, i.e. w.r.t. the producer's
 * BN output: masks by the activation, scatters through max-pool, and accumulates the two
 * per-channel BN-backward sums (sum g, sum g*z) */
enum { RD_DST_PLAIN = 0, RD_DST_POOL = 1, RD_DST_UPY = 2, RD_DST_NONE = 3 };

typedef struct {
    void* g;             /* gradient tensor to write, NHWC [*][Hd][Wd][Cd]                          */
    const void* z;       /* producer's raw tensor (PLAIN/POOL: same dims as g; UPY: t_lo, half res) */
    const float* scale;  /* producer BN scale/shift [G][Cd] (mask needs the sign of the BN output)  */
    const float* shift;
    double* bstats;      /* [G][RD_STAT_SLOTS][Cd][2] += (sum g, sum g*z)  or NULL                  */
    int32_t kind;
    int32_t act;         /* 1: an activation follows the producer BN                                */
    int32_t accumulate;  /* 1: g += (skip connection / second consumer)                             */
    int32_t Cd;
    float slope;
    int32_t n_off;       /* image offset into g / z                                                 */
    int32_t g_fixed;     /* >=0: producer group row                                                 */
    int32_t pad_;
} rd_dst_t;

/* rd_conv: replaces F.conv2d forward (unet.py:37-43,81-88,124-131,281,307) and, with flipped /
 * transposed packed weights, cudnn's dgrad; fused: bias add, BN batch statistics of the output
 * (nn.BatchNorm2d train mode), and on the read side the producer's BN+act+pool/upsample+concat. */
typedef struct {
    rd_src_t src[2];     /* 2 sources = torch.cat([prev, y], 1) (unet.py:110), never materialised */
    int32_t nsrc;
    int32_t taps;        /* 9 (3x3, pad 1) or 1 (1x1) */
    const void* w;       /* packed weights [CinPad/CK][taps][CoutPad][CK] (CK = 64 B of channels), dtype, zero padded */
    const float* bias;   /* [Cout] or NULL */
    int32_t CinPad, CoutPad;
    int32_t N, H, W, Cin, Cout;
    int32_t G;
    int32_t gstart[RD_MAX_GROUPS_C + 1];
    int32_t emode;       /* 0 forward: write out (+stats); 1 gradient: write through dst[] */
    void* out;           /* forward: NHWC [N][H][W][Cout] */
    double* stats;       /* forward: [G][RD_STAT_SLOTS][Cout][2] += (sum, sum of squares) of the result BEFORE the bias, or NULL */
    rd_dst_t dst[2];     /* gradient: channels [0,c_split) -> dst[0], [c_split,Cout) -> dst[1] */
    int32_t c_split;
    int32_t cu_limit;    /* >0: compute-unit budget of the launch where the kernel is persistent (the small-channel kernel, the
                          * warp-specialised 64-wide kernel): a conv of a side lane leaves the rest of the GPU to the main
                          * chain; 0: the whole device */
    int32_t w_tap_rows;  /* rows per tap of the packed array `w` points into when that is MORE than CoutPad: a launch over a 32-row
                          * block of a wider pack (w = pack + first_row * CK elements; honoured by the small-channel kernels, which is
                          * where such launches go); 0: CoutPad */
    int32_t stat_slots;  /* >0: `stats` and dst[].bstats are spread over the first stat_slots copies only (RD_STAT_SLOTS_FOLD); 0: all.
                            0..RD_STAT_SLOTS (else -2), and never MORE than the nslots of the consumer that finalizes these sums */
} rd_conv_t;

int rd_conv(const rd_conv_t* p, int dtype, void* stream);
/* 1 if rd_conv(p, dtype) will write the src[i].out tensors it is given (a pure function of the descriptor and the device) */
int rd_conv_honours_src_out(const rd_conv_t* p, int dtype);

/* rd_wgrad: replaces cudnn's wgrad for the same convs.  partial is a caller workspace of
 * rd_wgrad_workspace() bytes; the result is accumulated (beta=1) or stored (beta=0) into dW, fp32
 * OIHW [Cout][Cin][kh][kw]. */
typedef struct {
    rd_src_t a[2];       /* the conv's forward input, described exactly as for rd_conv */
    int32_t na;
    int32_t taps;
    rd_src_t dz;         /* gradient w.r.t. the conv output (RD_SRC_BNBWD or RD_SRC_RAW) */
    int32_t N, H, W, Cin, Cout;
    int32_t G;
    int32_t gstart[RD_MAX_GROUPS_C + 1];
    float* partial;
    float* dW;
    float beta;
    int32_t cu_limit;    /* >0: workgroup budget of the (persistent) launch in compute units -- a launch that runs on a side
                          * stream beside the dgrad chain leaves the rest of the GPU to it; 0: the whole device.  Enters
                          * rd_wgrad_workspace() (one partial per workgroup).  The 16-channel layers ignore it. */
} rd_wgrad_t;

int64_t rd_wgrad_workspace(const rd_wgrad_t* p, int dtype);
int rd_wgrad(const rd_wgrad_t* p, int dtype, void* stream);

/* Backward of a small-channel 3x3 conv (<= 32 channels in and out, bf16, plain per-pixel sources and destinations: every
 * 400x400 / 200x200 layer of the U-Net) in ONE launch: cudnn's dgrad AND wgrad behind the same nn.Conv2d
 * (unet.py:37-43,81-88,124-131,281,307).  Both halves are HBM-bound and read the same tensors -- g, z and the producer's raw
 * tensor; fused they are read once, and the weight-gradient MFMAs run on matrix pipes the dgrad leaves idle.
 * `dgrad` is the rd_conv descriptor of the gradient launch (emode 1), `wgrad` the rd_wgrad descriptor of the same conv;
 * rd_conv_bwd_fused_ok() says whether the pair qualifies (else use rd_conv + rd_wgrad).  rd_conv_bwd_fused writes the input
 * gradient exactly as rd_conv would and one block of weight-gradient sums per workgroup into wgrad->partial
 * (rd_conv_bwd_fused_workspace() bytes; wgrad->cu_limit is ignored, dgrad->cu_limit sizes the launch);
 * rd_conv_bwd_fused_reduce then sums them into wgrad->dW in a fixed order (a separate call so that a caller can run it
 * on another stream than the dgrad chain). */
int rd_conv_bwd_fused_ok(const rd_conv_t* dgrad, const rd_wgrad_t* wgrad, int dtype);
int64_t rd_conv_bwd_fused_workspace(const rd_conv_t* dgrad, const rd_wgrad_t* wgrad, int dtype);
int rd_conv_bwd_fused(const rd_conv_t* dgrad, const rd_wgrad_t* wgrad, int dtype, void* stream);
int rd_conv_bwd_fused_reduce(const rd_conv_t* dgrad, const rd_wgrad_t* wgrad, int dtype, void* stream);

/* rd_pack_weights: OIHW fp32 master weights -> the packed operand rd_conv reads.
 * K-chunk-major, CK = 32 (bf16) / 16 (fp32) input channels = the 64 bytes one K step of rd_conv consumes:
 * transpose=0: forward  [CinPad/CK][tap][CoutPad][CK]  (tap = kh*3+kw)
 * transpose=1: dgrad    [CoutPad'/CK][tap'][CinPad'][CK] with tap' = 8-tap (180-degree flip), i.e. the conv
 *              that maps dz (Cout channels) to da (Cin channels). */
int rd_pack_weights(const float* w_oihw, void* packed, int Cout, int Cin, int taps, int transpose,
                    int dtype, void* stream);
int64_t rd_packed_elems(int Cout, int Cin, int taps, int transpose, int dtype);

/* every conv of the network in ONE launch (run after each optimizer step): entry e reads the OIHW
 * weight at params+src_off and owns packed elements [start, next start) at packed+dst_off. */
typedef struct {
    int64_t src_off;     /* floats into the parameter arena */
    int64_t dst_off;     /* elements into the packed arena (== start) */
    int64_t start;
    int32_t Cout, Cin, taps, transpose, RowPad, ColPad;
} rd_pack_entry_t;
int rd_pack_weights_batched(const float* params, void* packed, const rd_pack_entry_t* table_dev, int n_entries,
                            int64_t total, int dtype, void* stream);


/* ------------------------------------------------------------------------------------------------
 * BatchNorm2d (train: batch statistics; eval: running statistics) split into "finalize" launches
 * around the convs.  nn.BatchNorm2d / DomainSpecificBatchNorm2d: code/networks/unet.py:17-28,
 * code/networks/dsbn.py:24-27.  Per-group parameter pointers: the two passes of the encoder share one
 * BN (same pointers in both groups -> running stats updated twice, in order); DSBN gives each domain
 * group its own bns[d]. */
typedef struct {
    const double* stats;  /* [G][RD_STAT_SLOTS][C][2] sum, sum of squares of (conv output - conv_bias) */
    const float* conv_bias; /* [C] bias of the producing conv (its sums exclude it) or NULL            */
    float* scale;         /* [G][C] out: gamma*invstd          */
    float* shift;         /* [G][C] out: beta - mean*scale     */
    float* mean;          /* [G][C] out (saved for backward)   */
    float* invstd;        /* [G][C] out                        */
    const float* gamma[RD_MAX_GROUPS_C];
    const float* beta[RD_MAX_GROUPS_C];
    float* running_mean[RD_MAX_GROUPS_C];
    float* running_var[RD_MAX_GROUPS_C];
    int64_t* num_batches_tracked[RD_MAX_GROUPS_C];
    float count[RD_MAX_GROUPS_C];   /* elements per channel in the group: N_g*H*W */
    int32_t C, G;
    float eps, momentum;
    int32_t training;     /* 0: eval -- scale/shift from the running statistics, nothing updated */
    int32_t nslots;       /* rd_src_t.fin only: statistic copies the producers used (0 = all); a multiple of 8 up to RD_STAT_SLOTS (else -3),
                             >= every producer's stat_slots.  rd_bn_finalize_fwd sums all RD_STAT_SLOTS copies */
} rd_bn_fwd_t;
int rd_bn_finalize_fwd(const rd_bn_fwd_t* p, void* stream);

/* BatchNorm backward, second half: from (sum g, sum g*z) produce the per-channel coefficients
 * dz = P*g + Q*z + R that the next dgrad / wgrad fold into their reads, and accumulate dgamma, dbeta. */
typedef struct {
    const double* bstats; /* [G][RD_STAT_SLOTS][C][2] */
    const float* mean;    /* [G][C] */
    const float* invstd;  /* [G][C] */
    const float* gamma[RD_MAX_GROUPS_C];
    float* dgamma[RD_MAX_GROUPS_C];   /* += */
    float* dbeta[RD_MAX_GROUPS_C];    /* += */
    float* P; float* Q; float* R;     /* [G][C] out */
    float count[RD_MAX_GROUPS_C];
    int32_t C, G;
    /* rd_gn_finalize_bwd only (NULL / ignored for BatchNorm, where the bias of a conv in front of the normalisation has an identically
     * zero gradient): GroupNorm pools the mean over the channels, so the conv bias has a gradient, sum over the pixels of dz =
     * P * sum g + Q * sum z + R * count, formed here from the forward sums */
    const double* fstats;   /* [G][RD_STAT_SLOTS][C][2] the forward statistics of this site */
    const float* conv_bias; /* [C] as in rd_bn_fwd_t (NULL: the forward sums already contain it) */
    float* dbias;           /* [C] += */
    int32_t nslots;         /* rd_src_t.fin only, as in rd_bn_fwd_t */
    int32_t pad_;
} rd_bn_bwd_t;
int rd_bn_finalize_bwd(const rd_bn_bwd_t* p, void* stream);

/* nn.GroupNorm(1, planes) (code/networks/unet.py:20-21, normalization(norm='gn')): every image is its own group (gstart = 0, 1, .., N;
 * N <= RD_MAX_GROUPS_C), mean / variance pooled over (C, H, W) of the image; the descriptors and outputs are those of the BatchNorm
 * finalize launches above (running-statistics pointers are ignored, `training` is irrelevant: the statistics are always the input's),
 * gamma / beta / dgamma / dbeta of group 0 are the layer's (shared by all images; dgamma / dbeta += in image order).
 * nn.InstanceNorm2d (unet.py:22-23, norm='in') needs no entry point of its own: rd_bn_finalize_fwd / _bwd with one group per image,
 * gamma -> ones, beta -> zeros, running_* = num_batches_tracked = dgamma = dbeta = NULL, training = 1. */
int rd_gn_finalize_fwd(const rd_bn_fwd_t* p, void* stream);
int rd_gn_finalize_bwd(const rd_bn_bwd_t* p, void* stream);

/* statistics of y = bilinear_x2(t) (nn.Upsample(scale_factor=2, 'bilinear', align_corners=False),
 * unet.py:84): stats[G][slot][C][2] += (sum y, sum y^2) (each thread sums deviations from the first value it sees and
 * converts to plain sums in fp64).  t: NHWC [N][h][w][C].  y_out == NULL: y is not
 * materialised (its readers interpolate t on the fly, RD_SRC_UP / RD_DST_UPY); otherwise y is also written to
 * y_out NHWC [N][2h][2w][C] in `dtype` and the statistics are those of the STORED (rounded) values, so that
 * plain RD_SRC_AFFACT / RD_DST_PLAIN readers normalise exactly what was measured. */
int rd_up_stats(const void* t, double* stats, void* y_out, int N, int h, int w, int C, int G, const int32_t* gstart_host,
                int dtype, int stat_slots /* 0: all copies */, void* stream);

/* statistics of a plain NHWC tensor x [N][H][W][C]: stats[G][slot][C][2] += (sum x, sum x^2) -- the standalone
 * nn.BatchNorm2d / DomainSpecificBatchNorm2d.forward (code/networks/dsbn.py:24-27), where no conv produced x. */
int rd_bn_stats(const void* x, double* stats, int N, int H, int W, int C, int G, const int32_t* gstart_host, int dtype,
                void* stream);

/* backward of y = up2(t) followed by BN: dt = up2^T( P*g + Q*up2(t) + R ), g: NHWC [N][2h][2w][C],
 * dt/t: NHWC [N][h][w][C]. */
int rd_up_bwd(const void* g, const void* t, void* dt, const float* P, const float* Q, const float* R, int N, int h,
              int w, int C, int G, const int32_t* gstart_host, int dtype, const rd_bn_bwd_t* fin /* NULL, or as rd_src_t.fin: produces P, Q, R */,
              int fin_flags, void* stream);

/* nn.MaxPool2d(2) at the head of ConvD levels 2-5 (unet.py:45,56), materialised: out[N][Ho][Wo][C] = max over the 2x2 window of
 * act(z*scale+shift) (scale == NULL: identity coefficients; slope 1: no activation), z: NHWC [N][2Ho][2Wo][C].  The conv behind
 * it then reads `out` as a plain tensor in forward, dgrad and wgrad. */
int rd_pool_fwd(const void* z, const float* scale, const float* shift, float slope, void* out, int N, int Ho, int Wo, int C, int G,
                const int32_t* gstart_host, int dtype, const rd_bn_fwd_t* fin /* NULL, or as rd_src_t.fin: produces scale, shift */, int fin_flags,
                void* stream);
/* its backward: g[N][2Ho][2Wo][C] (+)= gp scattered to the FIRST maximum of each window (ATen's tie rule), times the
 * activation derivative there when act != 0; bstats (or NULL) += (sum g_new, sum g_new*z) of the scattered values. */
int rd_pool_bwd(const void* gp, const void* z, const float* scale, const float* shift, float slope, int act, void* g, int accumulate,
                double* bstats, int N, int Ho, int Wo, int C, int G, const int32_t* gstart_host, int dtype,
                int stat_slots /* 0: all copies */, void* stream);

/* Elementwise materialisation  out = act( a*x + b*x2 + c ; slope )  over NHWC [N][H][W][C] with per-(group, channel)
 * coefficients a, b, c [G][C] (b, x2 may be NULL; slope 1 = no activation):
 *   forward  : a = BN scale, c = BN shift          -> the activated tensor relu(bn(z)) (unet.py:59-70), stored once
 *   backward : a = P, x2 = z, b = Q, c = R         -> dz, the gradient w.r.t. the conv output behind the BN
 * Used for the >= 64-channel layers, whose conv kernels are instruction-bound in the loader: their many readers
 * (one workgroup per 64 output channels, forward + dgrad + wgrad) then fetch the stored tensor with RD_SRC_RAW
 * instead of each redoing the BN/ReLU arithmetic; the tensors are small (<= 20 MB) at those depths. */
int rd_bn_apply(const void* x, const void* x2, void* out, const float* a, const float* b, const float* c, float slope,
                int N, int H, int W, int C, int G, const int32_t* gstart_host, int dtype, void* stream);

/* layout / materialisation at the module boundary (torch NCHW fp32 <-> NHWC dtype) */
/* cstride: elements between pixels of y (0 or C: dense; larger: zero-initialised pad channels are left untouched) */
int rd_nchw_to_nhwc(const float* x_nchw, void* y_nhwc, int N, int C, int H, int W, int cstride, int dtype, void* stream);
/* y_nchw = act(z*scale+shift)  (act: 0 none, 1 activation with `slope`); scale==NULL -> identity */
int rd_nhwc_to_nchw(const void* z_nhwc, float* y_nchw, const float* scale, const float* shift, int act, float slope,
                    int N, int C, int H, int W, int G, const int32_t* gstart_host, int dtype, void* stream);
/* gradient entering a module from torch: g_nhwc (+)= mask(dy_nchw), bstats += (sum g, sum g*z) */
int rd_grad_in(const float* dy_nchw, const void* z_nhwc, void* g_nhwc, const float* scale, const float* shift,
               double* bstats, int act, float slope, int accumulate, int N, int C, int H, int W, int G,
               const int32_t* gstart_host, int dtype, void* stream);
/* column sums of an NHWC tensor: out[c] (+)= sum over pixels (bias gradient of the two out1 convs) */
int rd_colsum(const void* x_nhwc, float* out, float* partial_ws /* >= 8192 floats */, int64_t npix, int C, int cstride /* 0: C */,
              float beta, int dtype, void* stream);


/* ------------------------------------------------------------------------------------------------
 * Fused segmentation + consistency losses, forward and backward in two passes over the logits.
 * Fundus  (code/train.py:246-259,283): p=sigmoid(l); BCELoss + dice_loss (utils/losses.py:8-16) for
 *          both passes + 0.5*KD|MSE(p2,p1) (train.py:85-88).  mask: NCHW fp32 [B][K][H][W].
 * Prostate (train.py:412-427,451): p=softmax(l); CrossEntropyLoss + dice_loss_multi(ignore_index=0)
 *          (utils/losses.py:18-33) + 0.5*KD|MSE.  target: int64 [B][H][W].
 * logits/dlogits: NHWC [2B][H][W][K] (pass 0 = images [0,B), pass 1 = [B,2B)).
 * losses_out[8] = {seg1, dice1, seg2, dice2, consistency, seg1+seg2+dice1+dice2+w*consistency, 0, 0}. */
typedef struct {
    const void* logits;
    const void* target;     /* fundus: const float* mask NCHW; prostate: const int64_t* */
    void* dlogits;
    float* partial;         /* workspace: rd_seg_loss_workspace() bytes */
    float* losses_out;      /* [8] device */
    int32_t B, H, W, K;
    int32_t kind;           /* 0 fundus (sigmoid/BCE/dice), 1 prostate (softmax/CE/dice_multi) */
    int32_t consistency;    /* 0 none, 1 kd, 2 mse */
    float cons_weight;      /* 0.5 (train.py:283) */
    int32_t dlogits_cstride; /* elements between pixels of dlogits (0 or K: dense); the pad is never written */
} rd_seg_loss_t;
int64_t rd_seg_loss_workspace(const rd_seg_loss_t* p);
int rd_seg_loss(const rd_seg_loss_t* p, int dtype, void* stream);

/* Restoration loss (train.py:265-276): per domain group d, MSELoss(tanh(rec_logits[d]), img[d]);
 * loss += lambda_rec * mse_d.  rec_logits / target / dlogits: NHWC [B][H][W][C].  mse_out[G] device.  H*W*C < 2^24 (-1 otherwise). */
int rd_rec_loss(const void* rec_logits, const void* target, void* dlogits, float* mse_out, float* partial_ws,
                int B, int H, int W, int C, int target_cstride /* 0: C */, int dlogits_cstride /* 0: C */, int G,
                const int32_t* gstart_host, float lambda_rec, int dtype, void* stream);
int64_t rd_rec_loss_workspace(int B, int H, int W, int C);

/* ------------------------------------------------------------------------------------------------
 * Adam over the flat parameter arena (torch.optim.Adam, betas (0.9,0.999), eps 1e-8, no weight decay;
 * code/train.py:573-576) with the reference's 3 param groups (encoder at lr/2) and poly LR
 * lr*(1-iter/total)^0.9 written after the step (train.py:289-293).  `iter` is a device counter that
 * the call increments, so a captured hipGraph replays the schedule without host involvement.
 * hyper_out[4] = {lr used for group 1/2, bias_correction1, bias_correction2, iter used}. */
typedef struct {
    float* param; const float* grad; float* exp_avg; float* exp_avg_sq;
    int64_t n;              /* total elements */
    int64_t n_half_lr;      /* the first n_half_lr elements (encoder) use lr/2 */
    int32_t* iter;          /* device, incremented */
    float* hyper_out;       /* device [4] */
    float base_lr;
    int32_t total_iters;
    float beta1, beta2, eps;
    int32_t pad_;
} rd_adam_t;
int rd_adam_step(const rd_adam_t* p, void* stream);

/* optimizer.zero_grad() (code/train.py:285,454) and the per-step reset of the BatchNorm sum buffers (stats / bstats arenas,
 * which the conv epilogues add into): n device ranges [ptrs_host[i], +bytes_host[i]) := 0, asynchronously on `stream`.
 * The two arrays are HOST arrays (read before the call returns).  Up to 8 ranges with 16-byte-aligned starts are cleared by ONE
 * kernel launch; otherwise one hipMemsetAsync per range. */
int rd_zero(void* const* ptrs_host, const int64_t* bytes_host, int n, void* stream);


/* ------------------------------------------------------------------------------------------------
 * Random Amplitude Mixup for a whole batch on the GPU.  Replaces extract_amp_spectrum /
 * low_freq_mutate_np / source_to_target_freq (code/dataset/fundus.py:13-61 == prostate.py:10-62) and
 * the call sites fundus.py:211-225 (clip [0,255], /127.5-1) and prostate.py:186-188 (clip [-1,1]).
 * src/trg: NHWC [B][H][W][3] (HWC like the PIL / .npy arrays the reference feeds the trio), fp32 or -- src_u8 != 0 --
 * uint8 (decoded PNG pixels: 1 byte per value to upload and to read); lam[B]: the per-sample ratio the reference draws
 * with random.randint(1,10)/10 (fundus.py:35); b = floor(0.1*min(H,W)) (fundus.py:26).
 * out_img = f(src), out_freq = f(clip(RAM(src,trg,lam), clip_lo, clip_hi)) with f(v) = v/div + offset when div != 0
 * (bit for bit the reference's `img /= 127.5; img -= 1.0`, fundus.py:217-218) else v*scale + offset; both NHWC `dtype`
 * [B][H][W][out_cstride] -- normally the two halves of the network input batch.
 * trg_amp != NULL: the partner is given as its amplitude spectrum [B][3][H][W] (what extract_amp_spectrum returns) instead
 * of as an image -- source_to_target_freq(src_img, amp_trg) of fundus.py:41; trg is then ignored.
 * tw_w / tw_h: (cos, -sin)(2*pi*k/N) tables of W / H entries.  H and W must factor into 2, 3 and 5 and be <= 1024; 256,
 * 384, 400 and 512 run compile-time radix plans. */
typedef struct {
    const void* src; const void* trg; const float* lam;
    void* out_img; void* out_freq;
    void* workspace;            /* rd_ram_workspace() bytes */
    const float* tw_w; const float* tw_h;
    int32_t B, H, W, C, b;
    float clip_lo, clip_hi, scale, offset;
    int32_t out_cstride;        /* elements between pixels of out_img / out_freq (0 or 3: dense; one 16-byte slot (8 bf16 /
                                 * 4 fp32): the first conv reads whole channel vectors, channels 3.. are written as zeros) */
    int32_t src_u8;             /* src / trg are uint8 */
    float div;
    int32_t pad_;
    const float* trg_amp;
    const void* dft_tables;     /* rd_ram_dft_tables() output for this (H, W, b), or NULL: see below */
} rd_ram_t;
int64_t rd_ram_workspace(int B, int H, int W, int b);
int rd_ram_mix(const rd_ram_t* p, int dtype, void* stream);
/* The mix only needs the bins |kx|, |ky| <= b of each spectrum (41 of 201 row bins at 400 x 400): with coefficient tables for the
 * geometry -- rd_ram_dft_tables_bytes() bytes of device memory filled ONCE by rd_ram_dft_tables (fp64 sincos, three bf16 terms per
 * coefficient) and passed as rd_ram_t.dft_tables -- rd_ram_mix computes exactly those bins as matrix products on the matrix cores
 * (csrc/ram_dft.hip) instead of whole FFTs on the vector units.  dft_tables == NULL, or a geometry the matrix path does not take
 * (rd_ram_dft_tables_bytes returns 0: W not a multiple of 16): the FFT kernels. */
int64_t rd_ram_dft_tables_bytes(int H, int W, int b);
int rd_ram_dft_tables(void* tables, int H, int W, int b, void* stream);

/* The reference's free functions on their own (API parity; the training step uses rd_ram_mix):
 *   rd_ram_amp     extract_amp_spectrum (fundus.py:13-19): amp[C][H][W] = |fft2(img[C][H][W])|, fp32
 *   rd_ram_mutate  low_freq_mutate_np (fundus.py:21-39) with the ratio given: out = amp_src, and inside the centred window
 *                  |ky| <= b, |kx| <= b (fftshift'ed coordinates c-b..c+b): lam*amp_src + (1-lam)*amp_trg */
int64_t rd_ram_amp_workspace(int C, int H, int W);
int rd_ram_amp(const float* img_chw, float* amp_chw, int C, int H, int W, void* workspace, const float* tw_w, const float* tw_h,
               void* stream);
int rd_ram_mutate(const float* amp_src, const float* amp_trg, float* out, int C, int H, int W, int b, float lam, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Launch lists.  The training step (code/train.py:225-296) is ~250 calls of the entry points above over three HIP streams; issued
 * one by one from the interpreter (a ctypes call + an event record / stream wait per fork) they cost the host 2.2 ms per 4.4 ms step.
 * rd_run_list walks a prepared list in C++: one call per step segment.
 *   op      which entry point (RD_OP_*), or RD_OP_FORK / RD_OP_JOIN;
 *   a[]     its arguments in declaration order WITHOUT the trailing stream: pointers and integers as they are, a float as its
 *           bit pattern in the low 32 bits.  Descriptor structs and host arrays (gstart_host, rd_zero's arrays) are referenced, not
 *           copied: they must stay alive and unchanged while the list is in use;
 *   lane    index into streams[] (0 = the caller's main stream).  An entry on lane k > 0 with wait_main != 0 first makes streams[k]
 *           wait for everything enqueued on streams[0] so far (the weight-gradient launches: each behind its place in the dgrad
 *           chain).  RD_OP_FORK: streams[lane] waits for streams[0]; RD_OP_JOIN: streams[0] waits for streams[lane] if the lane is
 *           open.  A lane is OPEN from its first fork / wait_main entry until it is joined.
 * *open_lanes (bit k = lane k; may be NULL) carries the open set in and out, so that consecutive segments can leave a lane running
 * across their boundary (the data-parallel step joins the weight-gradient lane only in front of the optimizer);
 * rd_join_lanes makes streams[0] wait for every lane in `mask`.  Works eagerly and under stream capture (events only).
 * Returns 0, the first failing entry point's error code, or -1 for a malformed entry (its index in *bad_index if not NULL). */
enum {
    RD_OP_FORK = 1, RD_OP_JOIN = 2,
    RD_OP_CONV = 10, RD_OP_WGRAD, RD_OP_CONV_BWD_FUSED, RD_OP_CONV_BWD_FUSED_REDUCE, RD_OP_PACK_WEIGHTS_BATCHED,
    RD_OP_BN_FINALIZE_FWD, RD_OP_BN_FINALIZE_BWD, RD_OP_GN_FINALIZE_FWD, RD_OP_GN_FINALIZE_BWD,
    RD_OP_UP_STATS, RD_OP_BN_STATS, RD_OP_UP_BWD, RD_OP_POOL_FWD, RD_OP_POOL_BWD, RD_OP_BN_APPLY,
    RD_OP_NCHW_TO_NHWC, RD_OP_NHWC_TO_NCHW, RD_OP_GRAD_IN, RD_OP_COLSUM,
    RD_OP_SEG_LOSS, RD_OP_REC_LOSS, RD_OP_ADAM_STEP, RD_OP_ZERO, RD_OP_RAM_MIX
};
#define RD_LAUNCH_MAX_ARGS 18
typedef struct {
    int32_t op;
    int32_t lane;
    int32_t wait_main;
    int32_t nargs;
    uint64_t a[RD_LAUNCH_MAX_ARGS];
} rd_launch_t;
int rd_run_list(const rd_launch_t* ops, int n, void* const* streams, int n_streams, uint32_t* open_lanes, int* bad_index);
int rd_join_lanes(void* const* streams, int n_streams, uint32_t mask);
/* enable != 0: rd_run_list enqueues the entries of every lane k > 0 from a worker thread of the library (one per lane, started on first
 * use), in parallel with the calling thread's lane 0: a launch costs the host ~2.4 us inside the HIP runtime whoever issues it, so ONE
 * thread needs ~0.85 ms for the step's ~305 kernels and three need ~0.4.  Same per-stream order, same events between the streams: the
 * GPU executes the same graph.  Not used while the main stream is being captured.  Returns the previous setting.  Process-wide;
 * rd_run_list itself must still be called from one thread at a time. */
int rd_run_list_threads(int enable);
/* enable != 0 (the default): a lane fork that directly follows a launch of the main stream takes its event from that launch's own
 * dispatch packet (hipExtLaunchKernelGGL stop event) instead of a hipEventRecord behind it: a record is a packet of its own between two
 * dependent kernels and costs the main stream ~3 us per fork (scripts/probe/ext_event.hip), ~40 times per training step.  The
 * dependency is the same one -- the lane waits for that launch and everything before it on the main stream.  Never under stream
 * capture, single-threaded walk only (an entry that launches no kernel of the library leaves the fork to a record).  Returns the previous setting.  Process-wide. */
int rd_run_list_bind_fork_events(int enable);
/* forks of the single-threaded walk since the library was loaded: served by a bound event / by a recorded one (tests, diagnostics) */
void rd_run_list_fork_counts(long long* bound, long long* recorded);
/* carries[i] = 1 for the entries of lane 0 that rd_run_list would launch with a fork's event bound to them (a RD_OP_FORK of a lane > 0,
 * or a lane entry with wait_main, follows before the main stream is given anything else or joins a lane).  Host logic only. */
int rd_run_list_fork_plan(const rd_launch_t* ops, int n, unsigned char* carries);

/* Measurement only (bench.py `box`): what this box's GPU sustains on two fixed micro-kernels, so that a bench line can be compared across
 * boxes of a pool whose clocks differ by a few per cent.  No reference counterpart (the reference publishes no throughput: BASELINE.md).
 *   which 0: streaming copy of n bytes (n % 16 == 0) from a to b, 16 B per lane, 8 workgroups of 256 threads per CU;
 *   which 1: n iterations of four independent v_mfma_f32_32x32x16_bf16 per wave, 8 waves on every CU (a: >= 4 bytes of device scratch, b unused):
 *            flops = CUs x 8 x n x 4 x 32768.
 *   which 2: stamps {shader-clock counter, 100 MHz counter} of every compute unit that one of 4096 small workgroups lands on, into
 *            a[2 * unit + 0..1] (2048 units x two 8-byte words, unit = XCC_ID << 8 | SE_ID << 5 | SH_ID << 4 | CU_ID; untouched slots
 *            keep their contents), in stream order (b, n unused): two launches around a stretch of work on one stream give its
 *            average shader clock per unit = 100 MHz x d(a[2u]) / d(a[2u + 1]) (counters of different XCDs are not aligned).
 * The caller times the launch with events on `stream`. */
int rd_box_probe(int which, void* a, void* b, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif
