/*
 * gdpso.h -- C ABI of libgdpso.so: the PSO latent-discovery hot path on MI355X (gfx950).
 *
 * The reference (ltronchin/Gan-discovery-PSO) is pure Python and has no FFI of its own;
 * each entry point below replaces the Python function(s) named in its comment
 * (paths relative to the reference root).  All signatures are plain C: pointers, sizes
 * and scalars.  Unless a parameter is marked HOST, pointers are device pointers on the
 * current HIP device; `stream` is a hipStream_t passed as void*.  Every call is
 * asynchronous on `stream` and never synchronises, allocates or frees device memory,
 * except the *_create / *_destroy functions (set-up time only).
 *
 * Return value: 0 on success, a GDPSO_ERR_* code otherwise; gdpso_last_error() gives the
 * message of the calling thread's most recent failure.
 *
 * Layouts: activations are NHWC fp32; swarm state is float64 (positions, velocities,
 * personal bests: the reference keeps them in numpy float64) with float32 fitness values
 * (the reference's fitness is a float32 array).
 */
#ifndef GDPSO_H
#define GDPSO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: gdpso_swarm gained cand_stride and the candidate records two distance-sum slots (GDPSO_CAND_DOUBLES);
 *    gdpso_swarm_scratch_bytes(n) grew about eightfold (one partial record per 32 particles instead of
 *    per 256): size the scratch buffer with the function, never with a constant from an older build
 * 3: gdpso_swarm_scratch_bytes(n) holds two sets of the distance diagnostic's tile sums (the fused loops alternate
 *    between them); gdpso_resident_options takes -1 for "the default" */
#define GDPSO_ABI_VERSION 3

#define GDPSO_OK 0
#define GDPSO_ERR_ARG 1       /* bad argument / unsupported shape */
#define GDPSO_ERR_HIP 2       /* a HIP runtime call failed */
#define GDPSO_ERR_WORKSPACE 3 /* caller's workspace is too small */

/* fitness modes: reference src/pso/util_discovery.py:75-78 */
#define GDPSO_MODE_IN_TRAINING 0  /* fitness = min(p + thr, 1) + eps        */
#define GDPSO_MODE_OUT_TRAINING 1 /* fitness = (1 - min(p + thr, 1)) + eps  */

/* Storage / matrix-core input type of a network handle: fp32 everywhere (the reference's
 * precision), or activations and convolution weights kept as bf16 / fp16 and multiplied by the
 * 16-bit matrix instructions, with fp32 accumulation and an fp32 epilogue (BatchNorm scale / shift,
 * residual add, activation).  The reference has no reduced-precision mode; images, logits and
 * fitness values cross the ABI as fp32 whatever the handle's type. */
#define GDPSO_DTYPE_F32 0
#define GDPSO_DTYPE_BF16 1
#define GDPSO_DTYPE_F16 2

int gdpso_abi_version(void);
const char *gdpso_last_error(void);
/* sizeof(gdpso_swarm) for which = 0, sizeof(gdpso_run) for which = 1, else 0: lets a foreign
 * binding check its struct layouts against the library's. */
size_t gdpso_struct_size(int which);

/* ------------------------------------------------------------------ generator
 * Replaces Generator.forward, reference src/utils/util_dcgan.py:128-149 (identical copies
 * src/inverter/utils_ae/util_inverter.py:119-140 and
 * src/inverter/utils_vq_vae/util_model.py:57-78, the VQ-VAE decoder used as generator),
 * followed by the per-image min/max rescale of src/utils/util_data.py:129-135 as called
 * from Discovery.particles_to_img, src/pso/util_discovery.py:33-50.
 */
typedef struct gdpso_generator gdpso_generator;

/* `tensors` (HOST): 14 fp32 arrays in state_dict order and PyTorch layout:
 *   gen.0.0.weight [Z,2F,k,k]  gen.0.0.bias [2F]
 *   gen.0.1.weight, .bias, .running_mean, .running_var [2F]
 *   gen.1.0.weight [2F,F,4,4]  gen.1.0.bias [F]
 *   gen.1.1.weight, .bias, .running_mean, .running_var [F]
 *   gen.2.weight [F,1,4,4]     gen.2.bias [1]
 * first_kernel k is 7 in the reference (28x28 images); output side = 4k. */
int gdpso_generator_create(gdpso_generator **out, int z_dim, int features_g, int first_kernel,
                           const float *const *tensors, int n_tensors);
/* The same with the element type of the activations between the layers and of the ConvTranspose
 * weights chosen by `dtype` (GDPSO_DTYPE_*; gdpso_generator_create is the GDPSO_DTYPE_F32 case).
 * The last layer (F -> 1 channel), tanh and the rescale are computed in fp32 either way.  The
 * 16-bit types need features_g to be a multiple of 64.  No counterpart in the reference (which is
 * fp32 throughout): results differ from it by the rounding of the stored activations. */
int gdpso_generator_create_typed(gdpso_generator **out, int z_dim, int features_g, int first_kernel,
                                 const float *const *tensors, int n_tensors, int dtype);
void gdpso_generator_destroy(gdpso_generator *g);
int gdpso_generator_image_side(const gdpso_generator *g);
size_t gdpso_generator_workspace_bytes(const gdpso_generator *g, int n);

/* z: [n, z_dim] latents, float64 if z_is_f64 else float32 (the reference casts float64
 * positions to float32 before the generator, util_discovery.py:41-43).
 * img   [n, S, S]: generator output in [-1,1]           (may be NULL)
 * img01 [n, S, S]: (img - min) / (max - min) per image
 * target [n, S, S] + mse [n]: optional (both or neither); mse[i] = mean((target_i - img_i)^2),
 * the reconstruction term of Discovery.fitness_pso_inverter, util_discovery.py:97. */
int gdpso_generator_forward(const gdpso_generator *g, void *stream, int n, const void *z, int z_is_f64,
                            float *img, float *img01, const float *target, float *mse,
                            void *workspace, size_t workspace_bytes);

/* ------------------------------------------------------------------- assessor
 * Replaces Resnet.forward + resnet_block.forward, reference src/pso/util_cnn.py:81-190,
 * built by get_cnn (:24-38), and the softmax/fitness tail of Discovery.fitness,
 * src/pso/util_discovery.py:66-80 (+ :84-101 for the inverter variant).
 */
typedef struct gdpso_assessor gdpso_assessor;

/* depths: blocks per stage ({3,4,6,3} = ResNet50, {3,4,23,3} = 101, {3,8,36,3} = 152).
 * Only 1 input channel is supported (the reference's data.channel).
 * `tensors` (HOST) fp32, PyTorch layout, in forward order:
 *   conv1.weight [64,1,7,7], bn1.{weight,bias,running_mean,running_var}
 *   then for every block: conv1.weight, bn1.x4, conv2.weight, bn2.x4, conv3.weight, bn3.x4,
 *        and, on the first block of each stage, identity_downsample.0.weight,
 *        identity_downsample.1.x4
 *   fc.weight [n_class,2048], fc.bias [n_class]
 * img_side: input image side the plan is specialised for (28, 64, ...). */
int gdpso_assessor_create_resnet(gdpso_assessor **out, const int depths[4], int n_class, int img_side,
                                 const float *const *tensors, int n_tensors);
/* The same with feature maps and convolution weights stored as `dtype` (GDPSO_DTYPE_*); the stem
 * (1 input channel), BatchNorm scale / shift, residual adds, pooling, classifier and softmax stay
 * fp32.  gdpso_assessor_create_resnet is the GDPSO_DTYPE_F32 case. */
int gdpso_assessor_create_resnet_typed(gdpso_assessor **out, const int depths[4], int n_class, int img_side,
                                       const float *const *tensors, int n_tensors, int dtype);
/* Replaces AlexNet.forward, reference src/pso/util_cnn.py:193-249 (get_cnn("AlexNet", ...), :25-26):
 * four ksize x ksize convolutions 1 -> 32 -> 64 -> 128 -> 256 with bias, each followed by the
 * activation and MaxPool2d(2); flatten; Linear(-> 256), Linear(256 -> 256), Linear(256 -> n_class)
 * with the activation after the first two (Dropout is inactive in eval mode).  pad_same: cnn_args
 * padding 'same' (else 'valid'); activation: 1 ReLU, 2 LeakyReLU(0.2) (cnn_args cnn_activation).
 * `tensors` (HOST) fp32, PyTorch layout: conv1..conv4 .weight/.bias, fc1..fc3 .weight/.bias (14).
 * Fails like the reference's constructor when a convolution's input is smaller than its window
 * (padding 'valid' at 28x28).  The handle is used through gdpso_assessor_forward like a ResNet's. */
int gdpso_assessor_create_alexnet(gdpso_assessor **out, int n_class, int img_side, int ksize, int pad_same, int activation,
                                  const float *const *tensors, int n_tensors, int dtype);
void gdpso_assessor_destroy(gdpso_assessor *a);
size_t gdpso_assessor_workspace_bytes(const gdpso_assessor *a, int n);
int gdpso_assessor_n_class(const gdpso_assessor *a);
/* nominal dense FLOPs of one image (2 per multiply-accumulate, padding taps included) */
double gdpso_assessor_flops_per_image(const gdpso_assessor *a);

/* img01 [n, S, S] -> logits [n, n_class] (may be NULL) and/or fitness [n] (may be NULL).
 * fitness = mode(p[class_index], threshold) + eps, p = softmax(logits); with two classes
 * index 1 is used whatever class_index says (util_discovery.py:71-74).
 * mse (may be NULL): when given, fitness = w_ass*fitness + w_rec*mse + eps
 * (util_discovery.py:86-101: eps is added a second time). */
int gdpso_assessor_forward(const gdpso_assessor *a, void *stream, int n, const float *img01,
                           float *logits, float *fitness, int class_index, int mode, float threshold,
                           float eps, const float *mse, float w_ass, float w_rec,
                           void *workspace, size_t workspace_bytes);

/* --------------------------------------------------------------------- encoder
 * Replaces Encoder.forward, reference src/inverter/utils_ae/util_inverter.py:164-184
 * (used once to seed particle positions, src/pso/util_pso.py:93-112).
 * `tensors` (HOST): enc.0.weight [F,1,4,4], enc.0.bias, enc.2.0.weight [2F,F,4,4],
 * enc.2.0.bias, enc.3.weight [Z,2F,7,7], enc.3.bias.  With n_tensors = 10 four more arrays
 * follow (weight, bias, running_mean, running_var of a BatchNorm2d(2F) after the second
 * convolution): the encoder of VectorQuantizedVAE_GAN, reference
 * src/inverter/utils_vq_vae/util_model.py:273-279,294-299. */
typedef struct gdpso_encoder gdpso_encoder;
int gdpso_encoder_create(gdpso_encoder **out, int enc_dim, int features_e, const float *const *tensors,
                         int n_tensors);
void gdpso_encoder_destroy(gdpso_encoder *e);
size_t gdpso_encoder_workspace_bytes(const gdpso_encoder *e, int n);
/* img [n,28,28] in [-1,1] -> latent [n, enc_dim] fp32 */
int gdpso_encoder_forward(const gdpso_encoder *e, void *stream, int n, const float *img, float *latent,
                          void *workspace, size_t workspace_bytes);

/* ----------------------------------------------------------------------- swarm
 * Replaces Particle / Swarm arithmetic, reference src/pso/util_pso.py:24-193.
 * All buffers are caller-owned device memory.
 */
typedef struct {
    int n;              /* particles held by this process                      */
    int d;              /* dim_space                                           */
    int first_index;    /* global index of local particle 0 (0 on one GPU)     */
    int n_ranks;        /* processes sharing the swarm (1 on one GPU)          */
    double *pos;        /* [n,d]                                               */
    double *vel;        /* [n,d]                                               */
    double *pbest_pos;  /* [n,d]                                               */
    float *pbest_val;   /* [n]   +inf before the first evaluation              */
    float *fitness;     /* [n]   written by the assessor                       */
    double *gbest_pos;  /* [d]                                                 */
    /* control block, >= GDPSO_CTRL_DOUBLES doubles, zero-filled then initialised by
     * gdpso_swarm_reset: [0] w_inertia  [1] gbest_val  [2] gbest global index
     * [3] number of entries in g_best_val  [4] stop iteration (0 = running)
     * [5] iterations completed */
    double *ctrl;
    float *g_best_val;        /* [max_iter+1] grows on improvement only (util_pso.py:143-149) */
    float *g_best_val_dummy;  /* [max_iter]   one entry per iteration (util_pso.py:151)       */
    double *mean_dist;        /* [max_iter]   mean pairwise distance (util_pso.py:76-86), may be NULL */
    double *hist_pos;         /* [max_iter+1, n, d] particle.history, may be NULL     */
    double *hist_vel;         /* [max_iter+1, n, d] particle.history_vel, may be NULL */
    double *cand;             /* n_ranks records {val, global idx, pos[d], distance sums[2]} of
                                 GDPSO_CAND_DOUBLES(d) doubles, cand_stride doubles apart: what the
                                 ranks all-gather once per iteration */
    void *scratch;            /* >= gdpso_swarm_scratch_bytes(n), n = the largest particle count
                                 passed to local_best / mean_distance */
    size_t scratch_bytes;
    int max_iter;
    int cand_stride;          /* doubles from one record of cand to the next; 0 = GDPSO_CAND_DOUBLES(d).
                                 Larger when a rank's positions lie behind its record in the same
                                 table (gdpso_swarm_distance_part_table) */
} gdpso_swarm;

/* doubles per candidate record of a swarm with dim_space d */
#define GDPSO_CAND_DOUBLES(d) ((d) + 4)

#define GDPSO_CTRL_DOUBLES 8
#define GDPSO_CTRL_W 0
#define GDPSO_CTRL_GBEST_VAL 1
#define GDPSO_CTRL_GBEST_IDX 2
#define GDPSO_CTRL_N_IMPROVE 3
#define GDPSO_CTRL_STOP_ITER 4
#define GDPSO_CTRL_ITERS_DONE 5

size_t gdpso_swarm_scratch_bytes(int n);

/* Swarm.__init__ + the bookkeeping of Particle.__init__ (util_pso.py:25-37, 54-70):
 * pbest_pos = pos, pbest_val = +inf, g_best_val = [inf], history[0] = pos / vel.
 * pos and vel must already hold the initial state. */
int gdpso_swarm_reset(const gdpso_swarm *s, void *stream, double w_inertia);

/* Swarm.update_personal_best compare-and-copy (util_pso.py:122-124) for the local
 * particles, then this rank's best candidate -> s->cand[rank_slot]. */
int gdpso_swarm_local_best(const gdpso_swarm *s, void *stream, int iteration, int rank_slot);

/* Swarm.update_global_best (util_pso.py:135-151) over the n_ranks candidate records in
 * s->cand (after the all-gather when n_ranks > 1), the inertia decay of :178-179 and the
 * early-stopping test of :186-188. */
int gdpso_swarm_global_best(const gdpso_swarm *s, void *stream, int iteration, int schedule_inertia,
                            int early_stopping, double tolerance);

/* Swarm.move_particles (util_pso.py:153-157; Particle.update_velocity :43-50 and
 * update_position :39-41).  rand [n,2] float64 = (r1, r2) per particle, or NULL to draw
 * them on the device from Philox4x32-10 keyed by (seed, global particle index, iteration).
 * f32_first_move: in the inverter variant the initial positions are float32 encoder outputs
 * (util_pso.py:106-110), so numpy evaluates the two attraction terms of iteration 1 in
 * float32; set it to reproduce that rounding (only iteration 1 is affected). */
int gdpso_swarm_move(const gdpso_swarm *s, void *stream, int iteration, double w_cogn, double w_soci,
                     const double *rand, uint64_t seed, int f32_first_move);

/* Swarm.mse (util_pso.py:76-86): mean Euclidean distance over all unordered pairs of
 * `pos_all` [n_all, d] -> s->mean_dist[iteration-1]. */
int gdpso_swarm_mean_distance(const gdpso_swarm *s, void *stream, int iteration, const double *pos_all,
                              int n_all);

/* The same diagnostic shared between ranks that all hold `pos_all`: the tiles of pairs are dealt
 * round-robin to the s->n_ranks ranks, `distance_part` leaves the sum of this rank's distances in
 * its own record cand[rank_slot] (slot d+2 for even iterations, d+3 for odd ones, so that the sums
 * of two consecutive iterations can be under way at once) to travel with the next all-gather of
 * the records, and once the records have been exchanged `distance_finish` adds the ranks' sums in
 * rank order and writes the mean to s->mean_dist[iteration-1]: the same value on every rank.  (The
 * reference's loop is one process, util_pso.py:182; this is its Swarm.mse cut into n_ranks equal
 * pieces of work.)
 * `distance_part_table` is `distance_part` for the layout in which the positions travel with the
 * records: rank r's n_all / n_ranks positions lie right behind its record (cand_stride >=
 * GDPSO_CAND_DOUBLES(d) + n_all / n_ranks * d; s->pos of rank r points at its own block), so one
 * all-gather of the table moves records and positions together and no second collective is needed. */
int gdpso_swarm_distance_part(const gdpso_swarm *s, void *stream, int iteration, const double *pos_all,
                              int n_all, int rank_slot);
int gdpso_swarm_distance_part_table(const gdpso_swarm *s, void *stream, int iteration, int n_all,
                                    int rank_slot);
int gdpso_swarm_distance_finish(const gdpso_swarm *s, void *stream, int iteration, int n_all);

/* Everything a rank does between the exchange of the candidate records and the next iteration, as
 * one call (one launch for swarms that one workgroup can step): distance_finish(diag_iteration)
 * when diag_iteration > 0, then swarm_global_best(iteration) and swarm_move(iteration), with the
 * arguments those take. */
int gdpso_swarm_after_exchange(const gdpso_swarm *s, void *stream, int iteration, int schedule_inertia,
                               int early_stopping, double tolerance, double w_cogn, double w_soci,
                               const double *rand, uint64_t seed, int f32_first_move, int diag_iteration,
                               int n_all);

/* One whole iteration on one GPU (n_ranks == 1): generator, rescale, assessor, fitness,
 * personal/global best, move, optional diagnostics -- the body of the while loop in
 * Swarm.optimize, util_pso.py:174-189.  `target` switches on the inverter variant. */
typedef struct {
    int class_index, mode;
    float threshold, eps;
    const float *target;  /* [n,S,S] or NULL */
    float w_ass, w_rec;
    double w_cogn, w_soci;
    int schedule_inertia, early_stopping;
    double tolerance;
    const double *rand;   /* [max_iter, n, 2] or NULL (device RNG) */
    uint64_t seed;
    int diagnostics;      /* compute mean pairwise distance each iteration */
    int f32_first_move;   /* see gdpso_swarm_move */
    float *img01;         /* [n,S,S] scratch for the rescaled images */
    float *img;           /* [n,S,S] scratch for the raw images      */
    float *mse;           /* [n] scratch (inverter variant)          */
    void *workspace;      /* >= max(generator, assessor workspace)   */
    size_t workspace_bytes;
} gdpso_run;

int gdpso_swarm_iterate(const gdpso_swarm *s, const gdpso_generator *g, const gdpso_assessor *a,
                        const gdpso_run *r, void *stream, int first_iteration, int n_iterations);

/* img01 -> fitness as gdpso_assessor_forward, with the class of interest given per image
 * (class_per_image: DEVICE int [n]): several swarms, each after its own class, in one batch. */
int gdpso_assessor_forward_classes(const gdpso_assessor *a, void *stream, int n, const float *img01, float *logits,
                                   float *fitness, const int *class_per_image, int mode, float threshold, float eps,
                                   const float *mse, float w_ass, float w_rec, void *workspace, size_t workspace_bytes);

/* Several independent swarms through ONE generator + assessor batch per iteration: the reference
 * walks its iid classes one after the other with a swarm each (src/training/pso_discovery.py:174-218,
 * 8 classes x 32 particles in configs/dcgan_mnist.yaml:17-24,146-156); batched, an iteration of all
 * of them costs one network pass over the concatenated particles.  Every swarm keeps its own global
 * best, inertia, draws, stopping state and max_iter (a swarm past its max_iter or stopped early is
 * skipped) and comes out exactly as gdpso_swarm_iterate would leave it given the same fitness values.
 * swarms / runs: HOST arrays of n_swarms descriptors.  Requirements: one dim_space; n_ranks == 1; the
 * swarms' pos and fitness buffers are consecutive slices of one buffer, in order (that buffer is the
 * evaluation batch); runs[k] gives swarm k its w_cogn / w_soci / schedule / stopping / rand / seed /
 * diagnostics; mode, threshold, eps are shared; runs[0] carries the shared img / img01 / workspace
 * (sized for all particles together).  class_per_image: DEVICE int [all particles], or NULL when the
 * assessor has two classes or there is one swarm (then runs[0].class_index).  Not for the
 * inverter variant (target must be NULL). */
int gdpso_swarm_iterate_multi(const gdpso_swarm *swarms, const gdpso_run *runs, int n_swarms, const gdpso_generator *g,
                              const gdpso_assessor *a, const int *class_per_image, void *stream, int first_iteration,
                              int n_iterations);

/* ------------------------------------------------------- vector quantisation (VQ-VAE)
 * The step after the PSO run in the reference's VQ-VAE workflow.  All buffers on the device,
 * row-major fp32 unless noted; `codebook` is [k, d] (nn.Embedding(K, D).weight).
 *
 * gdpso_vq_seed_codebook: rows [row0, row0+n) of the codebook = float32(positions [n, d],
 * float64): the last-iteration positions of one swarm, as gathered per class by
 * upload_pso_disentangled_space (reference src/training/vq_vae.py:35-57) and written into
 * the embedding by pso_weights (src/inverter/utils_vq_vae/util_model.py:49-54,125-132).
 *
 * gdpso_vq_nearest: replaces VectorQuantization.forward, reference
 * src/inverter/utils_vq_vae/util_function.py:6-23: distances (|c|^2 + |x|^2) - 2 x.c^T in
 * fp32, indices[i] (int64) = arg-min over the k codes, lowest index among equal distances, a
 * NaN distance taking precedence like torch.min.  Optional outputs: codes [n, d] =
 * codebook[indices] (VectorQuantizationStraightThrough.forward, :34-44) and best_dist [n].
 * d <= 4096.  workspace: gdpso_vq_workspace_bytes(k, d).
 *
 * gdpso_vq_lookup: codes[i] = codebook[indices[i]] (the embedding lookup of
 * VectorQuantizedVAE_GAN.decode, util_model.py:313-316); an index outside [0, k) gives a row
 * of NaN. */
size_t gdpso_vq_workspace_bytes(int k, int d);
int gdpso_vq_seed_codebook(void *stream, float *codebook, int k, int d, int row0, const double *positions, int n);
int gdpso_vq_nearest(void *stream, int n, int d, int k, const float *inputs, const float *codebook,
                     long long *indices, float *codes, float *best_dist, void *workspace, size_t workspace_bytes);
int gdpso_vq_lookup(void *stream, int n, int d, int k, const float *codebook, const long long *indices, float *codes);

/* No counterpart in the reference: gdpso_vq_nearest for callers that cannot allocate device memory
 * or hand over a workspace (code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc).  Same inputs, outputs and results as gdpso_vq_nearest; the transposed
 * codebook and its squared norms go to a static array inside the library instead of a caller
 * workspace, so the call allocates and frees nothing.  Limit: gdpso_vq_workspace_bytes(k, d) <=
 * GDPSO_VQ_STATIC_MAX_BYTES (e.g. K = 256 codes of D = 100: 103 KB), otherwise GDPSO_ERR_WORKSPACE.
 * Every GPU the library runs on holds its own copy of the array until the process ends.  All
 * calls on one GPU share that GPU's array: calls that overlap on one GPU (different streams or
 * threads) race; calls on one stream, or otherwise ordered, are safe.  Every call rewrites all of
 * the array it reads, so nothing has to be cleared between calls. */
#define GDPSO_VQ_STATIC_MAX_BYTES (1u << 20)
int gdpso_vq_nearest_static(void *stream, int n, int d, int k, const float *inputs, const float *codebook,
                            long long *indices, float *codes, float *best_dist);

/* -------------------------------------------------------------------- measurement
 * No counterpart in the reference: a hook for benchmarks.  Between begin and end, every
 * launch of the implicit-GEMM convolution kernel (generator, assessor and encoder
 * convolutions) is bracketed by HIP events on its own stream; end() waits for them and
 * returns the summed kernel time, the number of launches and their nominal FLOPs
 * (2 per multiply-accumulate, padding taps included).  Not thread-safe. */
int gdpso_conv_timing_begin(int max_launches);
int gdpso_conv_timing_end(double *total_ms, int *launches, double *flops);
/* Algorithmic bytes of the launches timed since the last gdpso_conv_timing_begin: input, weights,
 * output and residual of every launch, each counted once. */
double gdpso_conv_timing_bytes(void);
/* FLOPs those launches executed: like the nominal count, minus the taps of border pixels that
 * fall in the padding (they are not multiplied at all). */
double gdpso_conv_timing_executed_flops(void);

/* No counterpart in the reference: the launch plan of one ksize x ksize convolution layer, computed
 * on the host alone (no device needed; for tests and tuning).  transposed != 0 describes
 * ConvTranspose2d(k4, s2, p1) instead (ksize, stride, pad ignored).  Writes to `out` (at most
 * max_ints ints): nphase, tile, splits, hout, wout, then for every phase
 *   ntaps, off_y, off_x, hm, wm, in_y0, in_x0, followed by ntaps triples (dy, dx, weight slot).
 * Phase p covers the output pixels (my*out_mul + off_y, mx*out_mul + off_x), (my, mx) in hm x wm,
 * and reads input pixel (my*in_stride + in_y0 + dy, mx*in_stride + in_x0 + dx) for each tap.
 * Returns the number of ints written, or a negative error code. */
int gdpso_conv_plan_describe(int n_images, int hin, int cin, int cout, int ksize, int stride, int pad, int transposed,
                             int num_cus, int *out, int max_ints);

/* No counterpart in the reference: n / d the way the kernels compute it (multiply by a host-made
 * constant, shift; the chip has no integer divide), on the host, so that a test can compare it
 * with the exact quotient.  0 <= n < 2^31, d >= 1. */
unsigned gdpso_fast_div_check(unsigned n, unsigned d);

/* No counterpart in the reference: switches of the chained launch (every convolution of the ResNet
 * stages behind the first in ONE persistent launch whose workgroups hand intermediates to each other
 * inside an XCD's L2).  enabled = 0: launch plans made from now on run one launch per layer (A/B runs
 * and tests); drop_xcc_mask: workgroups on these XCDs (bits 0-7, not all) stand aside, which exercises the
 * path for an uneven deal of workgroups to XCDs (a test aid; results do not change); teams (0: keep,
 * 1, 2): teams of workgroups per XCD in plans made from now on (two teams each own a sixteenth of the
 * images and walk the layers independently; measured slower than one on MI355X, kept as an A/B switch).
 * Chains run on a device only while the library has seen a single stream there (a persistent grid holds
 * every CU while its workgroups wait for each other, so work on a second stream could keep it from becoming
 * resident; when a second stream appears the library drains the first once and launches per layer from then
 * on); every call also forgets the streams seen so far. */
int gdpso_chain_debug(int enabled, unsigned drop_xcc_mask, int teams);

/* No counterpart in the reference (one process, one batch size): options of the launch plans made from
 * now on (a handle keeps the plans it has already made, one per batch size it has seen).
 * plan_images > 0: tile shapes and split-K counts are chosen as if every batch held that many images,
 * whatever its size (0: every batch by its own size, the default).  The tile shape fixes the order in
 * which a row's partial sums over K are added, so with one plan_images an image gets the same bits in
 * any batch: a swarm sharded over ranks (each evaluating its block of particles) then reproduces the
 * unsharded run bit for bit when both plan for the same size, e.g. the ranks' block size.
 * splitk = 0: K is never split over workgroups (a chained launch never splits it; with 0 the launches
 * per layer add in the chain's order at every batch size).  Environment: GDPSO_PLAN_IMAGES,
 * GDPSO_PLAN_SPLITK give the initial values. */
int gdpso_plan_options(int plan_images, int splitk);

/* No counterpart in the reference: switches of the resident stages (the stem and the bottleneck blocks of the
 * first two ResNet stages, reference src/pso/util_cnn.py:85-95, 103-110, 147-157, in ONE launch whose workgroups
 * keep two images' maps in LDS; assessors on 28x28 images only, any element type).  Applies to assessors
 * created and launch plans made from now on.  stages = 0: stem and every stage run through the tiled kernels
 * (A/B runs and tests); 1: the stem and the first stage are resident; 2 (default): the second stage as well.
 * max_images / min_images > 0: the largest / smallest (planned, see gdpso_plan_options) batch that takes the
 * resident kernel; 0 keeps the current limit, -1 goes back to the default: no upper limit; below, 64 images for fp32
 * networks (smaller batches leave most CUs without a workgroup, and a workgroup is bound by its CU's matrix pipe) and 1
 * with 16-bit storage (there the tiled launches are bound by their fixed costs and one launch for fifteen wins at
 * every size).  A workgroup takes one image up to one image per CU, two beyond, and batches of more than two images
 * per CU run in rounds of two images a workgroup with a last round of at most one image per CU as a launch of its own
 * with one image a workgroup.  The second stage rides along from 160 images on in fp32, always with 16-bit storage.
 * The paths add a row's products in different orders and agree to fp32 rounding (to the element type's rounding for
 * 16-bit storage).  Environment: GDPSO_STAGE1_RESIDENT (0: none), GDPSO_STAGE2_RESIDENT (0: first stage only),
 * GDPSO_STAGE1_RESIDENT_MAX, GDPSO_STAGE1_RESIDENT_MIN, GDPSO_STAGE2_RESIDENT_MIN give the initial values. */
int gdpso_resident_options(int stages, int max_images, int min_images);

/* No counterpart in the reference (its generator is three torch modules, src/utils/util_dcgan.py:128-149): a
 * generator of the reference's shape (28x28 images, 64 features, first kernel 7) can run as ONE launch: a workgroup
 * takes one or two latents, keeps the 7x7x128 and 14x14x64 maps in LDS, streams the second layer's weights per output
 * phase and writes the image (and its [0,1] copy, and the inverter's mean squared error).  Latents of more than 16
 * dimensions keep their first layer as a GEMM launch in front.  mode = 0: the tiled kernels at every batch size;
 * 1 (default): the one launch where it is expected to be faster, within the batch limits of
 * gdpso_resident_options (16-bit storage: everywhere; fp32: where nearly every CU gets a workgroup, e.g. 256 or from
 * 512 images on with 256 CUs); 2: wherever those limits allow (tests).  Takes effect from the next forward call.
 * First and last layer add in the tiled kernels' order; the second layer's sums are ordered differently (fp32
 * rounding apart).  Environment: GDPSO_GEN_RESIDENT (0, 1, 2) gives the initial value. */
int gdpso_resident_generator_option(int mode);

/* No counterpart in the reference: what the planner of the resident launches decides for a batch planned as
 * n_images on a device of num_cus CUs, under the current options.  Host only (no GPU call): lets a test check the
 * choice without a device.  *stage_images: images a workgroup of the assessor's launch takes (0: the tiled kernels
 * and the chained launch); *tail_images: how many of the batch's last images form a last round that runs as a second
 * launch with one image a workgroup (0: none); *second_stage: 1 when the launch goes on through the second ResNet
 * stage of an assessor that has one within the launch's limits; *generator_images: as stage_images, for the
 * generator's launch with networks of element type dtype.  Any of the four may be NULL. */
int gdpso_resident_describe(int n_images, int num_cus, int dtype, int *stage_images, int *tail_images, int *second_stage,
                            int *generator_images);

/* No counterpart in the reference: a test aid.  Launch plans made from now on give every workgroup of the
 * resident launch this many images (1 or 2) whatever the batch size (within the limits of
 * gdpso_resident_options); 0: the planner chooses again.  The two forms give an image the same bits. */
int gdpso_resident_force_images(int images_per_workgroup);

/* No counterpart in the reference: the tile shape (64, 32 or 16, as in gdpso_conv_probe) a convolution layer
 * would run with as a step of a chained launch over n_images images with `teams` (1 or 2) teams per XCD, or 0
 * when it cannot be a step (fewer than 32 images in a share of the batch, or the layer alone is planned with
 * 128x128 tiles); negative on a bad argument.  Host only: lets a test check the choice without a GPU. */
int gdpso_chain_tile_describe(int n_images, int hin, int cin, int cout, int ksize, int stride, int pad, int num_cus, int teams,
                              int dtype);

/* No counterpart in the reference: chained launches since the library was loaded, and how many of them
 * gave up waiting for their workgroups or for a step (0 on a healthy device; such a launch ends with
 * wrong results instead of hanging, and gdpso_last_error() says what it waited for).  Synchronises
 * the device. */
int gdpso_chain_stats(unsigned *launches, unsigned *aborts);

/* No counterpart in the reference: times one ksize x ksize convolution layer in isolation
 * on random data (allocates and frees its own buffers; a development / tuning aid).
 * tile (0 = planner's choice; 128 = 128x128, 64 = 64x64, 32 = 32x64 with the K-tile split over two wave
 * pairs, 16 = 32x32 with the K-tile split over the four waves) and splits (0 = planner's choice) override
 * the launch plan. */
int gdpso_conv_probe(int n_images, int hin, int cin, int cout, int ksize, int stride, int pad, int tile,
                     int splits, int with_res, int reps, double *avg_us, int *tile_used, int *splits_used);

/* The same for a layer stored as `dtype` (GDPSO_DTYPE_*). */
int gdpso_conv_probe_typed(int n_images, int hin, int cin, int cout, int ksize, int stride, int pad, int tile,
                           int splits, int with_res, int reps, double *avg_us, int *tile_used, int *splits_used, int dtype);

/* No counterpart in the reference: runs ONE convolution layer on the caller's data (a test
 * aid; whole-network calls reach a tile shape or split count only by way of the planner).
 * weight / scale / shift are HOST arrays: weight in PyTorch layout, [cout][cin][k][k], or
 * [cin][cout][4][4] when `transposed` (= ConvTranspose2d k4 s2 p1; ksize/stride/pad ignored);
 * out = act(conv(in) * scale[c] + shift[c] + res), act 0 none / 1 ReLU / 2 LeakyReLU(0.2).
 * in / res (may be null) / out are DEVICE arrays in NHWC.  tile and splits as for
 * gdpso_conv_probe.  Allocates and frees its own device copies of the weights and the
 * split-K workspace, runs on the null stream and synchronises before returning. */
int gdpso_conv_layer_run(int n_images, int hin, int cin, int cout, int ksize, int stride, int pad, int transposed,
                         int act, int tile, int splits, const float *weight, const float *scale, const float *shift,
                         const float *in, const float *res, float *out, int *tile_used, int *splits_used);

/* The same with the layer's activations and weights stored as `dtype` (GDPSO_DTYPE_*): in / res are
 * rounded to that type on the way in, the output is widened back to fp32 (cin a multiple of 64 for
 * the 16-bit types). */
int gdpso_conv_layer_run_typed(int n_images, int hin, int cin, int cout, int ksize, int stride, int pad, int transposed,
                               int act, int tile, int splits, const float *weight, const float *scale, const float *shift,
                               const float *in, const float *res, float *out, int *tile_used, int *splits_used, int dtype);

#ifdef __cplusplus
}
#endif
#endif /* GDPSO_H */
