/*
 * mamdr_hip.h -- C ABI of libmamdr_hip.so: the MI355X (gfx950) hot path of MAMDR.
 *
 * The reference (RManLuo/MAMDR) has no FFI; its de-facto boundary is the handful
 * of Keras/TF calls through which the meta-learning wrappers touch numerics
 * (SURVEY.md section 8b).  Each entry point below names the reference interface
 * it replaces (file:line under /root/reference).  The reference-side binding a
 * maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions
 *  - plain C, no torch/HIP types in signatures; `stream` is a hipStream_t passed
 *    as void* (NULL = the null stream).
 *  - every pointer named d_* is a DEVICE pointer owned by the caller and must
 *    stay valid while bound; the library allocates only its private workspace.
 *  - return 0 on success, negative MAMDR_E* on failure; mamdr_last_error() gives
 *    the text (thread-local).  No internal threads; a context is not re-entrant.
 *  - all launches are asynchronous on the context's stream; nothing here
 *    synchronises the device except mamdr_profile_read().
 */
#ifndef MAMDR_HIP_H
#define MAMDR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MAMDR_ABI_VERSION 18

enum {
    MAMDR_OK = 0,
    MAMDR_EINVAL = -1,      /* bad argument / unsupported configuration */
    MAMDR_ESTATE = -2,      /* call out of order (state/tables/data not bound) */
    MAMDR_EHIP = -3,        /* a HIP runtime call failed */
    MAMDR_ENOTBUILT = -4    /* feature named by the reference but not built in this round */
};

/* tower kinds: run.py:37-47 + model_zoo/DeepCTR/deepctr.py:24-50 name registry */
enum { MAMDR_TOWER_MLP = 0, MAMDR_TOWER_DEEPFM = 1, MAMDR_TOWER_STAR = 2,
       /* deepctr WDL (deepctr.py:29-32): linear tables + DNN = DeepFM without the FM term; same segments */
       MAMDR_TOWER_WDL = 3,
       /* deepctr PNN (deepctr.py:44-46; use_inner, no outer product): the mlp tower on [user | item | domain | <u,i> <u,d>
          <i,d>] -- the three inner products feed three more rows of the first kernel (segment MAMDR_SEG_W0X).  Steps of
          up to 2,048 rows (every reference config has batch_size 1,024); larger batches: the generic-layer engine */
       MAMDR_TOWER_PNN = 4,
       /* deepctr NFM (deepctr.py:33-35): linear tables (as WDL) + DNN over the 128 bi-interaction columns u i + (u + i) d.
          Same segments as WDL, but segment W0 is deepctr's [128, 256] kernel: it occupies rows 256..383 of the mlp
          tower's W0 (the bi-interaction takes the domain field's place in the input tile; rows 0..255 stay zero and
          belong to no segment).  Steps of up to 2,048 rows, as PNN */
       MAMDR_TOWER_NFM = 5 };
/* data splits: utils/dataset.py:79-92 */
enum { MAMDR_SPLIT_TRAIN = 0, MAMDR_SPLIT_VAL = 1, MAMDR_SPLIT_TEST = 2 };
/* optimisers: deepctr.py:55 (Adam) / specific_base_model.py:120, base_model.py:69 (SGD finetune) */
enum { MAMDR_OPT_ADAM = 0, MAMDR_OPT_SGD = 1,
       /* no update: add d total_loss / d theta of the batch (dropout off) to the bound accumulator --
          the meta pass of first-order MAML, model_zoo/maml.py:107-109,196-229 */
       MAMDR_OPT_ACCUMULATE = 2 };
/* merged_method: model_zoo/specific_base_model.py:164-172 */
enum { MAMDR_MERGE_PLUS = 0, MAMDR_MERGE_TIMES = 1 };
/* segments of the flat trainable vector, in Keras trainable_weights order */
enum {
    MAMDR_SEG_USER_EMB = 0, MAMDR_SEG_ITEM_EMB = 1, MAMDR_SEG_DOMAIN_EMB = 2,
    MAMDR_SEG_W0 = 3, MAMDR_SEG_W1 = 4, MAMDR_SEG_W2 = 5,
    MAMDR_SEG_B0 = 6, MAMDR_SEG_B1 = 7, MAMDR_SEG_B2 = 8,
    MAMDR_SEG_WO = 9, MAMDR_SEG_GB = 10,
    /* DeepFM 1-d linear tables (deepctr get_linear_logit): the user / item ones are in the
       vector only when emb_trainable (they inherit the feature column's `trainable`), behind the
       embedding tables; the domain one follows the global bias */
    MAMDR_SEG_LIN_USER = 11, MAMDR_SEG_LIN_ITEM = 12, MAMDR_SEG_LIN_DOMAIN = 13,
    /* Star tower (model_zoo/Star/star_fcn.py:61-103, partitioned_norm.py:56-100): shared kernels / biases
       (meta parameters), then PartitionedNorm gamma / beta (shared [384], specific [D][384]), specific
       kernels [D][in][out] and biases [D][out].  W0..B2 report count 0 for this tower. */
    MAMDR_SEG_STAR_WS0 = 14, MAMDR_SEG_STAR_WS1 = 15, MAMDR_SEG_STAR_WS2 = 16,
    MAMDR_SEG_STAR_BS0 = 17, MAMDR_SEG_STAR_BS1 = 18, MAMDR_SEG_STAR_BS2 = 19,
    MAMDR_SEG_PN_GAMMA_SHARED = 20, MAMDR_SEG_PN_BETA_SHARED = 21,
    MAMDR_SEG_PN_GAMMA_SPEC = 22, MAMDR_SEG_PN_BETA_SPEC = 23,
    MAMDR_SEG_STAR_WD0 = 24, MAMDR_SEG_STAR_WD1 = 25, MAMDR_SEG_STAR_WD2 = 26,
    MAMDR_SEG_STAR_BD0 = 27, MAMDR_SEG_STAR_BD1 = 28, MAMDR_SEG_STAR_BD2 = 29,
    /* uncertainty weighting: `log_var` [D] (model_zoo/uncertainty_weight/weighted_loss.py:23-28), last segment */
    MAMDR_SEG_LOG_VAR = 30,
    /* PNN: rows 384..386 of deepctr's first DNN kernel [387, 256] (the inner products' rows), behind everything else */
    MAMDR_SEG_W0X = 31,
    MAMDR_SEG_COUNT = 32
};
/* kernels whose device time can be profiled (mamdr_profile_*) */
enum { MAMDR_KERNEL_FWD_BWD = 0, MAMDR_KERNEL_WGRAD = 1, MAMDR_KERNEL_UPDATE = 2,
       MAMDR_KERNEL_EVAL = 3, MAMDR_KERNEL_GATHER = 4, MAMDR_KERNEL_EMB_SWEEP = 5,
       /* every other launch of a training call, so that the slots add up to the whole step: k_pass_prep, k_emb_rows,
          k_emb_catchup, k_lin_sweep, k_star_catchup, and as ONE timed group each [k_star_stats + k_star_prep] and
          PartitionedNorm's backward [k_star_pnb_partial + _final + _apply (+ k_star_dm_final)] */
       MAMDR_KERNEL_AUX = 6,
       /* k_emb_flush: the lazy table Adam's replay of every lagging row (forced every 32 Adam steps) */
       MAMDR_KERNEL_FLUSH = 7, MAMDR_KERNEL_COUNT = 8 };

typedef struct mamdr_ctx mamdr_ctx;

/* Model / step configuration.  Replaces the `model` + `train` sections consumed by
 * DeepCTR.__init__/build_model (model_zoo/base_model.py:14-33,
 * model_zoo/DeepCTR/deepctr.py:20-61,95-136). */
typedef struct mamdr_config {
    int32_t abi_version;     /* MAMDR_ABI_VERSION */
    int32_t tower;           /* MAMDR_TOWER_* */
    int32_t n_user;          /* dataset.n_uid, utils/dataset.py:50-52 */
    int32_t n_item;          /* dataset.n_pid, utils/dataset.py:53-55 */
    int32_t n_domain;        /* utils/dataset.py:63-65 */
    int32_t emb_dim;         /* model.user_dim == item_dim == domain_dim (128) */
    int32_t hidden[3];       /* model.hidden_dim (256,128,64) */
    int32_t max_batch;       /* dataset.batch_size */
    int32_t emb_trainable;   /* train.emb_trainable: user/item tables join the trainable vector */
    float dropout;           /* model.dropout (rate) */
    float l2_emb;            /* deepctr.py:118 l2_reg_embedding = 1e-5 */
    float l2_linear;         /* DeepFM l2_reg_linear (deepctr default 1e-5); ignored by the mlp tower */
    float adam_beta1, adam_beta2, adam_eps; /* tf.train.AdamOptimizer defaults 0.9/0.999/1e-8 */
    int32_t uncertainty_weight; /* 1: training loss = mean(BCE) / var_d^2 + log var_d + regularisers with one trainable
                                   var per domain (model_zoo/uncertainty_weight/weighted_loss.py:30-43); evaluation is
                                   unweighted, as the reference evaluates the base model.  Every tower of mamdr_create but Star */
} mamdr_config;

const char* mamdr_last_error(void);
int mamdr_abi_version(void);
/* Every environment switch this build reads (library, Python host, bench.py, tools, tests), one per line:
 * "NAME\twho reads it\teffect\n"; a trailing '*' marks a name prefix.  Diagnostics only -- the reference is configured by
 * its JSON files alone (run.py:20-33) and the defaults are what the parity tests and bench.py run.  mamdr_create /
 * mamdr_graph_create report (stderr, once per process) any MAMDR_* name of the environment that is NOT in this table;
 * mamdr_env_unknown returns how many there were.  (ABI 18) */
const char* mamdr_env_switches(void);
int mamdr_env_unknown(void);

/* --- lifetime: replaces DeepCTR(dataset, config) / build_model + compile
 *     (model_zoo/DeepCTR/deepctr.py:20-61). */
int mamdr_create(const mamdr_config* cfg, void* stream, mamdr_ctx** out);
int mamdr_destroy(mamdr_ctx* ctx);

/* --- flat trainable vector ("meta parameters"): replaces
 *     MAML._get_model_meta_parms with meta_parms ["all"] (model_zoo/maml.py:153-179).
 *     mamdr_param_count = number of floats of the flat vector INCLUDING alignment
 *     padding (padding elements stay 0).  Layout per segment via mamdr_param_segment;
 *     a segment absent from the vector (frozen tables) reports count 0. */
int64_t mamdr_param_count(const mamdr_ctx* ctx);
/* Length of the META prefix of the flat vector: the tensors `MAML._get_model_meta_parms` selects
 * (model_zoo/maml.py:153-179).  Equals mamdr_param_count for the mlp / deepfm towers (meta_parms ["all"]);
 * for the Star tower it covers the tables, shared kernels and shared biases
 * (config/Taobao-10/star_taobao.json:37-41) -- theta / phi / merged vectors have this length and the
 * outer updates and weight assignment act on this prefix only. */
int64_t mamdr_meta_count(const mamdr_ctx* ctx);
/* Non-trainable model state (Star: PartitionedNorm moving mean / variance per domain and the
 * zero-debias slots of their moving averages, partitioned_norm.py:71-87,177-193): number of floats
 * (0 for the other towers) and binding of a caller-owned device buffer, initialised by the caller
 * (layout: mov_mean [D][384] = 0 | mov_var [D][384] = 1 | biased_mean = 0 | biased_var = 0 | steps [D] = 0). */
int64_t mamdr_aux_count(const mamdr_ctx* ctx);
int mamdr_bind_aux(mamdr_ctx* ctx, float* d_aux);
int mamdr_param_segment(const mamdr_ctx* ctx, int seg, int64_t* offset, int64_t* count);

/* Bind the live model state (caller-owned device memory, mamdr_param_count floats
 * each): weights, Adam m, Adam v.  Replaces the TF variables + optimizer slots
 * created at deepctr.py:55-60.  Adam slots are NOT reset by weight assignment
 * (SURVEY A.5); mamdr_optimizer_reset zeroes m, v and the step count. */
int mamdr_bind_state(mamdr_ctx* ctx, float* d_params, float* d_adam_m, float* d_adam_v);
int mamdr_optimizer_reset(mamdr_ctx* ctx);
/* Bind the meta-gradient accumulator (mamdr_param_count floats, caller-owned, caller zeroes it):
 * target of MAMDR_OPT_ACCUMULATE steps.  Replaces `self.accum_grads` (model_zoo/maml.py:202). */
int mamdr_bind_accumulator(mamdr_ctx* ctx, float* d_acc);
int64_t mamdr_optimizer_steps(const mamdr_ctx* ctx);
/* Restore the run's two host-side counters -- the Adam step count with its running beta powers (TF's `beta1_power` /
 * `beta2_power` slot variables, restored by tf.train.Saver with the optimizer: deepctr.py:55-60 creates them) and the
 * position of the dropout stream (mamdr_dropout_steps) -- e.g. when a run resumes from saved weights and slots written into
 * the bound vectors.  The live state is synchronised first (as mamdr_sync_tables); afterwards every table row counts as
 * current AT `optimizer_steps`.  The parity tests use it to start a pass from the oracle's state at that point
 * (tests/test_gpu_teacher.py).  (ABI 18) */
int mamdr_set_counters(mamdr_ctx* ctx, int64_t optimizer_steps, int64_t dropout_steps);
/* Trainable tables only (no-op otherwise).  tf.train.AdamOptimizer moves every table row every step
 * (deepctr.py:54-60 with l2_reg_embedding: regulariser gradient + decaying moments).  The library replays
 * those per-row steps lazily -- bit-identical to the per-step dense update -- so between two calls of
 * mamdr_train_steps rows that no batch touched may lag.  Call this before READING the bound weights /
 * Adam slots from outside the library or REPLACING them (K.batch_get_value / SetVarOp, maml.py:181-194;
 * mamdr_copy / mamdr_interp / ... on the bound vectors).  mamdr_eval_domain, mamdr_gather_rows,
 * mamdr_optimizer_reset, mamdr_bind_state and SGD / accumulate steps synchronise by themselves.
 * A non-null d_loss_out makes every step synchronise first (its regulariser term sums over all rows).
 * MAMDR_DENSE_ADAM=1 in the environment keeps the per-step dense sweep instead. */
int mamdr_sync_tables(mamdr_ctx* ctx);
/* Launches of the lazy table Adam's replay kernel (k_emb_flush) so far: all of them, or (forced_only != 0) only those
 * the flush period forced inside a training call.  Lets a parity test state how many flushes a run spanned
 * (tests/test_gpu_fullsize.py); no reference counterpart -- TF1's dense Adam has no such event. */
int64_t mamdr_table_flushes(const mamdr_ctx* ctx, int32_t forced_only);

/* Bind the frozen user / item tables (row-major [rows, emb_dim] fp32).  Replaces
 * DeepCTR.build_emb with a Constant initializer, trainable=False
 * (deepctr.py:104-116).  seg = MAMDR_SEG_USER_EMB or MAMDR_SEG_ITEM_EMB.  With
 * emb_trainable the tables live inside the flat vector and this is an error. */
int mamdr_bind_table(mamdr_ctx* ctx, int seg, const float* d_rows, int64_t n_rows);

/* Bind one domain's split: int32 uid/pid/domain columns and fp32 label column of
 * n rows, in file order.  Replaces get_dataset / make_csv_dataset + expand_dim
 * (utils/dataset.py:12-38,79-92): columns uid,pid,domain,label. */
int mamdr_bind_domain_data(mamdr_ctx* ctx, int domain, int split, const int32_t* d_uid,
                           const int32_t* d_pid, const int32_t* d_domain, const float* d_label,
                           int64_t n_rows);

/* Run n_steps consecutive inner optimisation steps ("domain-steps") on one
 * domain's TRAIN split, entirely device-side.  Replaces the loops
 *   for step in range(train_step): model.train_on_batch(train_iter)
 * (model_zoo/mamdr.py:85-86,96-97, domain_negotiation.py:71-72, reptile.py:69-70)
 * and model.fit(iter, steps_per_epoch=n) (mamdr.py:54, deepctr.py:76).
 *   d_perm     row order of this pass (the shuffled iterator, utils/dataset.py:27-37),
 *              n_rows int32, or NULL for file order
 *   first_step first batch index within the pass; batch s covers
 *              perm[s*batch .. min(n_rows,(s+1)*batch))  (final partial batch kept,
 *              utils/dataset.py:25)
 *   dropout_seed seed of the counter-based dropout stream (step index = the
 *              context's global inner-step counter)
 *   d_loss_out optional device array of n_steps floats: total loss per step
 *              (BCE mean + regularisers), what train_on_batch returns as `loss`. */
int mamdr_train_steps(mamdr_ctx* ctx, int domain, const int32_t* d_perm, int64_t first_step,
                      int64_t n_steps, int32_t batch, uint32_t dropout_seed, int32_t optimizer,
                      float lr, float* d_loss_out);
/* Same, over a pass of `pass_rows` positions only (d_perm then holds pass_rows row indices of the split;
 * -1 = the whole split).  Replaces the take/skip sub-datasets of the meta-train / meta-val split
 * (model_zoo/maml.py:300-330, mldg.py:309-325: `dataset.take(n_train)` / `dataset.skip(n_train)`), whose
 * final partial batch ends at the sub-dataset's end. */
int mamdr_train_steps_n(mamdr_ctx* ctx, int domain, const int32_t* d_perm, int64_t pass_rows, int64_t first_step,
                        int64_t n_steps, int32_t batch, uint32_t dropout_seed, int32_t optimizer, float lr,
                        float* d_loss_out);

/* Evaluate one domain's split with the live weights, dropout off.  Replaces
 * model.evaluate(data, steps=n_step) (model_zoo/base_model.py:131,
 * specific_base_model.py:84).
 *   d_loss_out  1 float: mean over batches of the batch-mean loss (+ regularisers)
 *   d_hist      2*501 uint32 (zeroed by this call): d_hist[c*501 + k] = number of
 *               rows with (label != 0) == c whose prediction exceeds exactly k of
 *               the 500 AUC thresholds (utils/auc.py:118-126).  The confusion
 *               counts of utils/metrics_utils.py:297-354 are suffix sums of it.
 *   d_pred_out  optional n_rows floats: predictions in file order. */
int mamdr_eval_domain(mamdr_ctx* ctx, int domain, int split, int32_t batch, float* d_loss_out,
                      uint32_t* d_hist, float* d_pred_out);

/* Standalone embedding gather (K1 of SURVEY 2.2): out[r] = [U[uid]|I[pid]|Dm[dom]]
 * for n rows of a bound split in d_perm order.  Same code path the step kernel
 * uses for its tiles; exported for parity tests and bandwidth measurement. */
int mamdr_gather_rows(mamdr_ctx* ctx, int domain, int split, const int32_t* d_perm,
                      int64_t first_row, int64_t n_rows, float* d_out);

/* --- outer (meta) updates on flat vectors; stateless, any stream.  Each op is
 *     evaluated with one fp32 rounding per arithmetic step in the reference's
 *     order (no FMA contraction) and matches its numpy result bit-for-bit.
 *
 * dst[i] += (a[i] - b[i]) * scale
 *   DN / Reptile: model_zoo/domain_negotiation.py:118-123, reptile.py:127-132
 *   MAMDR:        model_zoo/mamdr.py:173-180 (b = merged or dst itself) */
int mamdr_interp(float* d_dst, const float* d_a, const float* d_b, float scale, int64_t n,
                 void* stream);
/* average_meta_grad == "moving_mean": `K.moving_average_update(ag, g, 0.999)` (model_zoo/maml.py:219-220,
 * mldg.py:222-223, pcgrad.py:229-230) = TF 1.12's zero-debiased moving average of the accumulator variable:
 *   biased[i] -= (biased[i] - value[i]) * decay;   unbiased[i] -= unbiased[i] - biased[i] / denom
 * decay = float(1 - momentum); denom = 1 - (1 - decay)^local_step, formed by the caller (float32, local_step =
 * number of updates since the accumulator was created, this one included).  One rounding per operation. */
int mamdr_moving_average(float* d_unbiased, float* d_biased, const float* d_value, float decay, float denom,
                         int64_t n, void* stream);
/* dst[i] = theta[i] + phi[i]  or  theta[i] * phi[i]  (specific_base_model.py:164-172) */
int mamdr_merge(float* d_dst, const float* d_theta, const float* d_phi, int32_t mode, int64_t n,
                void* stream);
/* One Domain-Regularisation support step on flat vectors in a single pass (mamdr.py:103-105, then the next
 * support's assignment of the merged weights to the model, mamdr.py:74):
 *   phi[i] += (w[i] - merged[i]) * gamma;  merged[i] = theta[i] (+|*) phi[i];  if (assign_model) w[i] = merged[i]
 * bit-identical to mamdr_interp(phi, w, merged, gamma) + mamdr_merge(merged, theta, phi, mode) + mamdr_copy(w, merged). */
int mamdr_dr_advance(float* d_phi, float* d_w, float* d_merged, const float* d_theta, float gamma, int32_t mode,
                     int32_t assign_model, int64_t n, void* stream);
/* The same on the context's LIVE weights, range [meta_off, meta_off + n) of the bound vector (d_w = live + meta_off):
 * what a caller gets from mamdr_sync_tables(ctx) followed by mamdr_dr_advance(.., live + meta_off, ..) -- the library
 * brings the live state up to date itself, and a domain-table step the fused step path left pending is materialised
 * inside the same launch.  Same bits. */
int mamdr_dr_advance_live(mamdr_ctx* ctx, float* d_phi, float* d_merged, const float* d_theta, float gamma, int32_t mode,
                          int32_t assign_model, int64_t meta_off, int64_t n);
/* dst[i] = a[i] - b[i]   (mamdr.py:168-171) */
int mamdr_sub(float* d_dst, const float* d_a, const float* d_b, int64_t n, void* stream);
/* acc[i] += (a[i] - b[i]) [* shared[i]] / divisor   (reptile.py:134-137 with shared NULL,
 * divisor 1; mamdr.py:182-191) */
int mamdr_accumulate(float* d_acc, const float* d_a, const float* d_b, const float* d_shared,
                     float divisor, int64_t n, void* stream);
/* dst[i] += acc[i] / divisor * scale; acc[i] = 0   (mamdr.py:193-196; reptile.py:139-142
 * with divisor <= 0 meaning "no division") */
int mamdr_apply_accumulated(float* d_dst, float* d_acc, float divisor, float scale, int64_t n,
                            void* stream);
/* TF1 Adam on flat vectors with caller-owned slots: the OUTER optimiser of MAML
 * (`self.meta_optimizer.apply_gradients`, model_zoo/maml.py:201,215,236-243).
 *   alpha = lr * sqrt(1 - beta2_power) / (1 - beta1_power), powers AFTER this step's update
 *   m += (g - m)(1 - beta1); v += (g*g - v)(1 - beta2); p -= m * alpha / (sqrt(v) + eps);
 *   grad_scale multiplies g first (average_meta_grad = "mean", maml.py:208-210; 1 otherwise). */
int mamdr_adam_apply(float* d_p, float* d_m, float* d_v, const float* d_g, float grad_scale, float lr,
                     float beta1, float beta2, float eps, float beta1_power, float beta2_power, int64_t n,
                     void* stream);
/* dst[i] = src[i]: SetVarOp.__call__ / K.batch_get_value without the host round
 * trip (utils/tool.py:36-45, maml.py:189-194) */
int mamdr_copy(float* d_dst, const float* d_src, int64_t n, void* stream);
/* PCGrad projection of one auxiliary gradient onto the running gradient, both flat device vectors, in place.
 * Replaces `PCGrad.PCGrad(final_grads, current_grads, aux_grads)` with final_grads IS current_grads
 * (model_zoo/pcgrad.py:107-124,152-160).  The n_seg tensors are given as HOST arrays: element offset, number of
 * slices along the last axis (rows), slice length (cols <= 4096); at most 24 tensors.  Bit-identical to numpy. */
int mamdr_pcgrad_project(float* d_final, float* d_aux, const int64_t* h_offsets, const int64_t* h_rows,
                         const int32_t* h_cols, int32_t n_seg, void* stream);

/* --- host-side helper: tf.data shuffle(buffer_size) order of range(n)
 *     (utils/dataset.py:27-37), splitmix64-driven; writes n int32 to HOST memory. */
int mamdr_shuffle_perm(int64_t n, int64_t buffer_size, uint64_t seed, int32_t* h_out);
/* the same for every pass of an epoch in one call: pass k is a shuffle of range(h_n[k]) with seed h_seeds[k],
 * written at h_out + sum(h_n[0..k)) (one pinned staging buffer, one upload per epoch instead of one per
 * re-initialised iterator: mamdr.py:52-53,81-82,92-93). */
int mamdr_shuffle_perms(int32_t n_passes, const int64_t* h_n, int64_t buffer_size, const uint64_t* h_seeds,
                        int32_t* h_out);

/* --- profiling: per-kernel device time from HIP events on the context's stream.
 *     enable != 0 brackets every launch of the listed kernels with events.
 *     mamdr_profile_read synchronises the stream, returns the summed milliseconds
 *     and launch count since the last reset. */
int mamdr_profile_enable(mamdr_ctx* ctx, int32_t enable);
/* A HINT about the next calls of mamdr_train_steps(_n): they will run these passes -- (domain, permutation, rows of the
 * pass; h_pass_rows null or an entry < 0: the whole split) at batch size `batch` -- in this order.  Where a call would
 * resolve and gather its pass's rows itself (frozen tables on the k_wgrad_adam path: k_pass_prep, once per call) the
 * library gathers all of them in ONE launch now, and each of those calls finds its rows ready; it matches a call by
 * (domain, d_perm pointer, pass rows, batch), skipping entries in between, and forgets the hint when a call matches
 * none (that call then gathers as before).  The permutations and the bound columns must not change in between.  At
 * most 16 passes per hint (more: the first 16); everywhere else this is a no-op.  Same rows, same bits.  No reference
 * counterpart: the reference's tf.data iterator re-reads the csv files on every pass (utils/dataset.py:20-38).
 * mamdr_pregather_hits: how many calls found their pass gathered; mamdr_pregather_launches: how many hints led to a
 * gather launch (k_pass_prep_multi) -- both for tests and reports. */
int mamdr_pregather_passes(mamdr_ctx* ctx, int32_t n_passes, const int32_t* h_domains, const int32_t* const* h_d_perms,
                           const int64_t* h_pass_rows, int32_t batch);
int64_t mamdr_pregather_hits(const mamdr_ctx* ctx);
int64_t mamdr_pregather_launches(const mamdr_ctx* ctx);
/* which kernels a training step of `batch` rows launches (for reports; no reference counterpart):
 *   0  tower -> k_wgrad (split-K slabs) -> k_update
 *   1  [k_pass_prep once per call] tower -> k_wgrad_adam (weight gradients + optimiser step in one launch;
 *      MAMDR_KERNEL_WGRAD times it), the domain table's step applied by the next tower, k_dm_finish once per call
 *      (MAMDR_KERNEL_UPDATE times it) */
int mamdr_step_path(const mamdr_ctx* ctx, int32_t batch);
/* rows per tower workgroup from the next call on: 0 = the library's choice (4-row tiles while the grid fits the CUs in one
 * round: ONE chain of steps then has every CU busy), 4 or 16 forced.  16 is the choice of a context that SHARES the
 * device with other contexts on other streams (the lanes of mamdr_amd/parallel.py: 64 workgroups per 1,024-row step
 * leave the other CUs to the other lanes' launches; 4 lanes: 72 K instead of 59 K domain-steps/s,
 * profiles/r05_lanes_probe.txt).  Same arithmetic per element either way; the two tiles add the split-K partials of a
 * layer in different orders (rounding-level differences, each inside the parity bars).  No reference counterpart.
 * (ABI 17; the environment's MAMDR_TOWER_TILE sets the initial value.) */
int mamdr_set_tower_tile(mamdr_ctx* ctx, int32_t rows);
/* rows per tower workgroup of a training step of `batch` rows under the present choice: 4 or 16 (for reports and tests) */
int mamdr_tower_tile(const mamdr_ctx* ctx, int32_t batch);
/* training steps taken so far (any optimiser): the position of the counter-based dropout stream, which the mask of
 * step s is keyed on (dropout_seed, s).  A caller that replays the run elsewhere continues the stream from here. */
int64_t mamdr_dropout_steps(const mamdr_ctx* ctx);
int mamdr_profile_reset(mamdr_ctx* ctx);
int mamdr_profile_read(mamdr_ctx* ctx, int32_t kernel, double* total_ms, int64_t* launches);

/* ====================================================================================================
 * Towers built from generic dense layers: the reference's multi-task baselines
 * (model_zoo/DeepMTLCTR/deep_mtl_ctr.py).  `DeepMTLCTR.build_model` (:21-67) builds deepctr's
 * SharedBottom / MMOE / PLE with one binary task per domain and compiles, per domain, `Model(inputs, outputs[d])` on ONE
 * shared tf.train.AdamOptimizer; `train()` (:69-96) fits domain d's model on domain d's batches.  A handle of its own
 * (the hot path's context is specialised to the 384-256-128-64 tower); same conventions as above.
 * The flat vector lists, in order: the domain table, the experts every task mixes, then per task its own experts (PLE),
 * its gate DNN + gate kernel, its tower DNN, its output unit and global bias (tensor names: mamdr_graph_tensor_info).
 * Frozen user / item tables are bound through mamdr_graph_bind_table; with emb_trainable = 1 (the Amazon configs) they sit
 * at the head of the flat vector and take TF1's dense Adam step (every row, every step) like any other tensor. */
enum { MAMDR_GRAPH_SHARED_BOTTOM = 0,   /* deep_mtl_ctr.py:25-30  models.SharedBottom */
       MAMDR_GRAPH_MMOE = 1,            /* deep_mtl_ctr.py:31-38  models.MMOE */
       MAMDR_GRAPH_PLE = 2,             /* deep_mtl_ctr.py:39-49  models.PLE (num_levels = 1, as in every reference config) */
       /* single-output towers of the deepctr family on the same generic layers (model_zoo/DeepCTR/deepctr.py): ONE model
          serves every domain; flat vector = [user_emb item_emb (lin_user lin_item) if trainable] domain_emb W0 W1 W2 b0 b1 b2
          wo gb (lin_domain); hidden_dim in expert_hidden, tower / gate fields unused */
       MAMDR_GRAPH_NFM = 3,             /* deepctr.py:33-35  models.NFM: linear tables + DNN(BiInteractionPooling) */
       MAMDR_GRAPH_PNN = 4,             /* deepctr.py:44-46  models.PNN: DNN([fields | pairwise inner products]) */
       /* deepctr.py:41-43  models.CCPM: convolutions (6, 1) x 4 and (5, 1) x 4 over the field axis with tanh and k-max pooling
          (k = 1 with three fields) -> DNN + linear tables; conv tensors conv1_w [6][4] conv1_b conv2_w [4][4] conv2_b precede W0 */
       MAMDR_GRAPH_CCPM = 5,
       /* deepctr.py:37-40  models.AutoInt(att_head_num=4): three InteractingLayers (4 heads x 8, residual, relu) over the three
          fields beside the DNN, Dense(1) on [attention output 96 | DNN output] + linear tables; att<l>_w = [W_query | W_key |
          W_value | W_res] ([128][128], then [32][128] twice) precede W0, wo has 96 + hidden[-1] rows */
       MAMDR_GRAPH_AUTOINT = 6,
       /* round 5: the towers of the step kernels with ANY hidden_dim of 1..4 layers (widths multiples of 64) -- the step
        * kernels are built for [256, 128, 64]; deepctr.py:26-32,36-38 pass hidden_dim through as dnn_hidden_units.
        * Flat layout = oracle/tower.param_names: [tables | 1-d linear tables] domain_emb | W0.. | b0.. | wo | gb | lin_domain */
       MAMDR_GRAPH_MLP = 7,             /* deepctr.py:118-136 build_mlp: DNN(x) -> Dense(1) -> sigmoid */
       MAMDR_GRAPH_WDL = 8,             /* deepctr.py:29-32  models.WDL: linear tables + DNN(x) */
       MAMDR_GRAPH_DEEPFM = 9 };        /* deepctr.py:36-38  models.DeepFM: linear tables + FM second-order term + DNN(x) */
typedef struct mamdr_graph mamdr_graph;
typedef struct mamdr_graph_config {
    int32_t abi_version;        /* MAMDR_ABI_VERSION */
    int32_t kind;               /* MAMDR_GRAPH_* */
    int32_t n_user, n_item, n_domain, emb_dim, max_batch, emb_trainable;
    int32_t n_expert_hidden; int32_t expert_hidden[4];   /* model.hidden_dim: bottom / expert DNN (deep_mtl_ctr.py:26,33,44) */
    int32_t n_tower_hidden;  int32_t tower_hidden[4];    /* model.tower_hidden_dim (:27,34,43) */
    int32_t n_gate_hidden;   int32_t gate_hidden[4];     /* model.gate_dnn_hidden_units (:36,45); unused by shared_bottom */
    int32_t num_experts;                                 /* mmoe (:32) */
    int32_t shared_expert_num, specific_expert_num;      /* ple (:40-41) */
    float dropout, l2_emb, adam_beta1, adam_beta2, adam_eps;
    float l2_linear;                                     /* NFM: deepctr l2_reg_linear (1e-5) on the 1-d linear tables */
    int32_t uncertainty_weight;                          /* single-output towers: the weighted loss of uncertainty_weight/
                                                          * weighted_loss.py:30-43 (one trainable `log_var` per domain, last
                                                          * tensor of the flat vector); evaluation stays unweighted */
} mamdr_graph_config;
const char* mamdr_graph_last_error(void);
int mamdr_graph_create(const mamdr_graph_config* cfg, void* stream, mamdr_graph** out);   /* build_model, deep_mtl_ctr.py:21-67 */
int mamdr_graph_destroy(mamdr_graph* g);
int64_t mamdr_graph_param_count(const mamdr_graph* g);
int32_t mamdr_graph_tensor_count(const mamdr_graph* g);
/* tensor i of the flat vector: Keras-style name ("expert_0/W1", "gate_3/Wg", "tower_3/b0", "head_3/w", ...), element offset
 * and shape (biases and scalars: rows = 1) -- model.trainable_weights of deep_mtl_ctr.py:51 */
int mamdr_graph_tensor_info(const mamdr_graph* g, int32_t i, char* name, int32_t name_cap, int64_t* offset, int64_t* rows,
                            int64_t* cols);
/* the two ranges of the flat vector a step on `domain` trains = trainable_weights of Model(inputs, outputs[domain])
 * (deep_mtl_ctr.py:62-66): the shared block and that task's block; everything else neither moves nor decays */
int mamdr_graph_task_ranges(const mamdr_graph* g, int domain, int64_t* shared_off, int64_t* shared_count, int64_t* task_off,
                            int64_t* task_count);
int mamdr_graph_bind_state(mamdr_graph* g, float* d_params, float* d_adam_m, float* d_adam_v);
int mamdr_graph_optimizer_reset(mamdr_graph* g);
/* epsilon of the following Adam steps: `compile(optimizer='adam')` of deep_mtl_ctr.py:147-148 builds a Keras Adam
 * (epsilon = K.epsilon() = 1e-7) where the shared optimiser of :53-56 is tf.train.AdamOptimizer (1e-8) */
int mamdr_graph_set_adam_eps(mamdr_graph* g, float eps);
int64_t mamdr_graph_optimizer_steps(const mamdr_graph* g);
/* kernel launches this process has issued through mamdr_graph_* calls so far (measurement: launches per step) */
int64_t mamdr_graph_launch_count(void);
int64_t mamdr_graph_dropout_steps(const mamdr_graph* g);
/* as mamdr_set_counters: the optimizer's step count (with TF's running beta powers) and the dropout stream's position (ABI 18) */
int mamdr_graph_set_counters(mamdr_graph* g, int64_t optimizer_steps, int64_t dropout_steps);
int mamdr_graph_bind_table(mamdr_graph* g, int seg, const float* d_rows, int64_t n_rows);     /* deep_mtl_ctr.py:98-121 */
int mamdr_graph_bind_domain_data(mamdr_graph* g, int domain, int split, const int32_t* d_uid, const int32_t* d_pid,
                                 const int32_t* d_domain, const float* d_label, int64_t n_rows);
/* n_steps x `domain_model_dict[domain]` train_on_batch: deep_mtl_ctr.py:79-80 (Adam) / :158-172 (per-domain fit).
 * Same batch / permutation / dropout-stream conventions as mamdr_train_steps; MAMDR_OPT_ADAM, MAMDR_OPT_SGD or
 * MAMDR_OPT_ACCUMULATE (after mamdr_graph_bind_accumulator). */
int mamdr_graph_train_steps(mamdr_graph* g, int domain, const int32_t* d_perm, int64_t first_step, int64_t n_steps,
                            int32_t batch, uint32_t dropout_seed, int32_t optimizer, float lr, float* d_loss_out);
/* as mamdr_train_steps_n: the pass covers `pass_rows` positions, d_perm lists that many rows of the split (the take / skip
 * sub-datasets of the meta-train / meta-val split, maml.py:300-330, mldg.py:309-325); pass_rows < 0 = the whole split */
int mamdr_graph_train_steps_n(mamdr_graph* g, int domain, const int32_t* d_perm, int64_t pass_rows, int64_t first_step,
                              int64_t n_steps, int32_t batch, uint32_t dropout_seed, int32_t optimizer, float lr,
                              float* d_loss_out);
/* MAMDR_OPT_ACCUMULATE steps of the meta wrappers on these towers (maml.py:107-109,196-229; mldg.py; pcgrad.py): dropout off,
 * no update, the step's gradient (every tensor the task's model trains, tables included) added to this flat vector of
 * mamdr_graph_param_count floats; NULL unbinds */
int mamdr_graph_bind_accumulator(mamdr_graph* g, float* d_acc);
/* `domain_model_dict[domain].evaluate(data, steps=n_step)`: deep_mtl_ctr.py:207; outputs as mamdr_eval_domain */
int mamdr_graph_eval_domain(mamdr_graph* g, int domain, int split, int32_t batch, float* d_loss_out, uint32_t* d_hist,
                            float* d_pred_out);

#ifdef __cplusplus
}
#endif
#endif /* MAMDR_HIP_H */
