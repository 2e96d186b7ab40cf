// CPU check of aspire_amd/csrc/asmc_bisect.h (test infrastructure): the round planner of the device-side temperature
// search against the reference's sequential bisection (src/aspire/samplers/smc/base.py:167-186) on the SAME function
// ESS(beta)/N - evaluated by direct double sums on synthetic populations - so every decision is the same comparison and
// beta* must agree bit for bit.  Prints one line per family with the distribution of rounds; exit code 1 on a mismatch.
//   g++ -O2 -std=c++17 -I aspire_amd/csrc tests/tools/bisect_plan_check.cpp -o bisect_plan_check
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "asmc_bisect.h"

static uint64_t rng_state = 88172645463325252ULL;
static double urand() {
    rng_state ^= rng_state << 13, rng_state ^= rng_state >> 7, rng_state ^= rng_state << 17;
    return (double)(rng_state >> 11) * (1.0 / 9007199254740992.0);
}
static double nrand() { return sqrt(-2.0 * log(urand() + 1e-300)) * cos(6.283185307179586 * urand()); }

struct Pop {
    std::vector<double> delta;
    double dmax;
    double beta0;
    long evals = 0;
    // ESS/N of the weights exp((beta - beta0) Delta_i)
    double eff(double beta, double* S1 = nullptr, double* S2 = nullptr) {
        evals++;
        const double t = beta - beta0;
        double s1 = 0.0, s2 = 0.0;
        for (double d : delta) {
            const double e = d == -INFINITY ? (t > 0.0 ? 0.0 : 1.0) : exp(t * (d - dmax));
            s1 += e, s2 += e * e;
        }
        if (S1) *S1 = s1, *S2 = s2;
        return s1 * s1 / s2 / (double)delta.size();
    }
};

static double sequential(Pop& p, double target, double tol) {  // smc/base.py:167-186
    double beta_min = p.beta0, beta_max = 1.0;
    if (p.eff(1.0) >= target) beta_min = 1.0;
    while (beta_max - beta_min > tol) {
        const double beta_try = 0.5 * (beta_max + beta_min);
        if (p.eff(beta_try) >= target)
            beta_min = beta_try;
        else
            beta_max = beta_try;
    }
    return beta_min;
}

struct Result {
    double beta;
    int rounds, windows, misses;
    bool trip_ok;
    double trip_s1;
};

static Result planned(Pop& p, double target, double tol, bool plain_only) {
    double st[40];
    memset(st, 0, sizeof(st));
    st[BIS_BMIN] = p.beta0, st[BIS_BMAX] = 1.0, st[BIS_TARGET] = target, st[BIS_TOL] = tol, st[BIS_BETA0] = p.beta0;
    st[BIS_N] = (double)p.delta.size(), st[BIS_M_ONE] = p.dmax * (1.0 - p.beta0);
    double beta[16], m[16], eff[16], y[16], S[32];
    for (int round = 0; round < 64; round++) {
        const int LU = round ? (int)st[BIS_LU] : 4;
        const long long Kf = round ? (long long)st[BIS_KFIRST] : 1, stride = round ? (long long)st[BIS_STRIDE] : 1;
        for (int j = 0; j < 16; j++) {
            beta[j] = bis_node_beta(Kf + j * stride, LU, p.beta0);
            if (Kf + j * stride > (1LL << LU)) beta[j] = 1.0 + (double)(Kf + j * stride - (1LL << LU)) * ldexp(1.0 - p.beta0, -LU);
            const int c = bis_col_of_sorted(j);
            eff[j] = p.eff(beta[j], &S[2 * c], &S[2 * c + 1]);
            m[j] = st[BIS_M_ONE] * ((beta[j] - p.beta0) / (1.0 - p.beta0));
            y[j] = log(eff[j]) - log(target);
        }
        if (plain_only) st[BIS_MODE] = 1.0;
        bis_plan(st, round == 0, beta, m, eff, y, S);
        if (st[BIS_DONE] != 0.0) break;
    }
    return {st[BIS_BMIN], (int)st[BIS_ROUNDS], (int)st[BIS_WINDOWS], (int)st[BIS_MISSES], st[BIS_TRIP_OK] != 0.0,
            st[BIS_TRIP_S1]};
}

int main(int argc, char** argv) {
    const int trials = argc > 1 ? atoi(argv[1]) : 300;
    const char* names[] = {"gaussian-like", "heavy tails", "peaked (Delta ~ 1e3)", "nearly uniform", "zero-likelihood rows",
                           "two clusters", "late stage (beta0 ~ 1)", "tolerance = a cell width"};
    int bad = 0;
    for (int kind = 0; kind < 8; kind++) {
        int hist[16] = {0}, hist_plain[16] = {0}, windows = 0, misses = 0, cases = 0;
        for (int trial = 0; trial < trials; trial++) {
            Pop p;
            const int n = 3 + (int)(urand() * urand() * 3000);
            p.delta.resize(n);
            const double b0s[] = {0.0, 0.0, 0.013, 0.4, 0.93, 0.999};
            p.beta0 = b0s[(int)(urand() * 6)];
            double tol = urand() < 0.5 ? 1e-6 : 1e-8;
            if (urand() < 0.1) tol = urand() < 0.5 ? 0.3 : 1e-3;
            const double targets[] = {0.5, 0.3, 0.9, 0.01};
            double target = targets[(int)(urand() * 4)];
            for (int i = 0; i < n; i++) {
                double d;
                switch (kind) {
                case 0: d = -0.5 * 8 * (1 + 0.5 * nrand()) * (1 + 0.5 * nrand()); break;
                case 1: { const double z = nrand(), w = fabs(nrand()) + 1e-3; d = 20.0 * z / w; break; }
                case 2: d = -fabs(nrand()) * 3e3; break;
                case 3: d = 1e-3 * nrand(); break;
                case 4: d = urand() < 0.2 ? -INFINITY : nrand() * 3; break;
                case 5: d = (urand() < 0.3 ? 40.0 : 0.0) + nrand(); break;
                case 6: d = nrand() * 50; p.beta0 = urand() < 0.5 ? 0.93 : 0.999; break;
                default: d = nrand() * 4 * (1 + urand()); break;
                }
                p.delta[i] = d;
            }
            p.dmax = -INFINITY;
            for (double d : p.delta) p.dmax = fmax(p.dmax, d);
            if (kind == 7) {  // a tolerance that IS the float width of some cell of the stopping level (other cells differ by an ulp)
                const int lev = 8 + (int)(urand() * 14);
                const long long k = (long long)(urand() * (double)(1LL << lev));
                tol = bis_node_beta(k + 1, lev, p.beta0) - bis_node_beta(k, lev, p.beta0);
                if (urand() < 0.3) tol = nextafter(tol, 0.0);
            }
            const double want = sequential(p, target, tol);
            const Result got = planned(p, target, tol, false), plain = planned(p, target, tol, true);
            cases++;
            hist[got.rounds < 15 ? got.rounds : 15]++, hist_plain[plain.rounds < 15 ? plain.rounds : 15]++;
            windows += got.windows, misses += got.misses;
            bool ok = got.beta == want && plain.beta == want;
            if (ok && want > p.beta0 && want < 1.0) {
                double s1, s2;
                p.eff(want, &s1, &s2);
                ok = got.trip_ok && got.trip_s1 == s1;
            }
            if (got.rounds > plain.rounds + 1) ok = false;  // a miss costs one round at most
            if (!ok) {
                bad++;
                if (bad < 20)
                    printf("MISMATCH kind %d trial %d n %d beta0 %.17g tol %.17g target %g: sequential %.17g planned %.17g (%d rounds) "
                           "plain %.17g (%d rounds) trip %d\n", kind, trial, n, p.beta0, tol, target, want, got.beta, got.rounds,
                           plain.beta, plain.rounds, (int)got.trip_ok);
            }
        }
        printf("%-26s %4d cases  rounds:", names[kind], cases);
        for (int r = 1; r < 16; r++)
            if (hist[r]) printf(" %d:%d", r, hist[r]);
        printf("   plain:");
        for (int r = 1; r < 16; r++)
            if (hist_plain[r]) printf(" %d:%d", r, hist_plain[r]);
        printf("   windows %d (missed %d)\n", windows, misses);
    }
    printf(bad ? "FAILED: %d mismatches\n" : "all equal to the sequential loop\n", bad);
    return bad ? 1 : 0;
}
