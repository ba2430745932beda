"""The oracle-ensemble criterion of the end-to-end and full-size AUC tests, shared by
tests/test_gpu_e2e.py and tests/test_gpu_fullsize.py; checked on synthetic draws by tests/test_teacher_harness.py."""
import numpy as np

TWIN_SEEDS = (99, 100, 101, 102, 103)          # K = 5 perturbed oracle twins of an `ensemble` case (the first one: every case)


class Ensemble(object):
    """K + 1 oracle runs of one case -- the unperturbed twin and K twins whose initial weights differ by one fp32 rounding
    (relative 2e-7, K different draws): what ANY fp32 evaluation of this training is distributed like.
    For every comparison c = (evaluation, domain) -- an epoch's validation, a test evaluation, the returned scores --
        mean_c  = mean of the K + 1 members' AUC
        loo_k,c = |member k - mean of the OTHER members|          (a member's distance from the ensemble without it)
        hip_c   = |hip - mean_c|
    and per run two statistics over all comparisons of the case: S = the MEAN distance (a systematic offset shows here) and
    M = the LARGEST distance (a single outlying domain shows here).  The HIP run must not be an outlier of the ensemble in
    either: each of its statistics lies inside the one-sided 99.9 % PREDICTION INTERVAL for one more member,
    mean_k + t(0.999; K) * sd_k * sqrt(1 + 1 / (K + 1)) (Student t with K degrees of freedom over the K + 1 members' values:
    6.4 sd for six members -- with two dozen such checks in the suite a run that IS a member fails one of them in ~2 % of
    the sessions; at 3 sd it would in ~30 %).
    (Any rank criterion -- "no further out than the furthest member" -- fails a run that IS a member with probability
    1 / (K + 2) by symmetry, whatever K is affordable; tests/test_teacher_harness.py checks these bars on synthetic draws.)
    Where north_star's plain |hip - oracle| <= 1e-3 holds nothing else is needed; the count of comparisons beyond it is
    printed next to each twin's own count against the same oracle run.  No factor on a single draw, no allowed share of misses."""

    def __init__(self, members):
        self.members = members            # summaries: oracle first, then the twins
        self.h_dist, self.m_dist = [], [[] for _ in members]
        self.n_cmp = self.beyond = 0
        self.m_beyond = [0] * (len(members) - 1)
        self.worst = 0.0

    def check(self, what, vals_h, vals_members):
        """vals_h: {d: AUC} of the HIP run; vals_members: the same per member."""
        K1 = len(vals_members)
        doms = sorted(vals_members[0])
        arr = np.array([[vm[d] for d in doms] for vm in vals_members], np.float64)          # [K + 1][D]
        mean = arr.mean(axis=0)
        loo = np.abs(arr - (arr.sum(axis=0, keepdims=True) - arr) / (K1 - 1))
        for j, d in enumerate(doms):
            diff = abs(vals_h[d] - vals_members[0][d])
            self.h_dist.append(abs(vals_h[d] - mean[j]))
            for k in range(K1):
                self.m_dist[k].append(loo[k, j])
            for k in range(1, K1):
                self.m_beyond[k - 1] += int(abs(arr[k, j] - arr[0, j]) > 1e-3)
            self.n_cmp += 1
            self.worst = max(self.worst, diff)
            self.beyond += int(diff > 1e-3)

    def aggregate(self, case):
        out = {}
        for name, f in (("mean", np.mean), ("largest", np.max)):
            s_h = float(f(self.h_dist))
            s_k = np.array([float(f(m)) for m in self.m_dist])
            from scipy import stats as _st
            n = len(s_k)
            width = float(_st.t.ppf(0.999, n - 1)) * np.sqrt(1.0 + 1.0 / n)
            out[name] = (s_h, s_k, float(s_k.mean() + width * s_k.std(ddof=1)), width)
        print("  ensemble of %d oracle runs, %d comparisons: |hip - oracle| worst %.1e, %d beyond the plain 1e-3 (the twins against the "
              "same oracle run: %s)" % (len(self.members), self.n_cmp, self.worst, self.beyond, self.m_beyond))
        for name in ("mean", "largest"):
            s_h, s_k, bar, width = out[name]
            print("    %s distance from the ensemble mean: hip %.2e | members (leave-one-out) %s | 99.9 %% prediction bound mean + %.1f sd = %.2e" % (
                name, s_h, " ".join("%.2e" % v for v in s_k), width, bar))
        for name in ("mean", "largest"):
            s_h, s_k, bar, width = out[name]
            assert s_h <= bar, ("the HIP run is an outlier of the oracle ensemble: %s distance" % name, case, s_h, list(s_k), bar)
