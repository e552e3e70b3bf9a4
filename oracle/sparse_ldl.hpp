// ORACLE (test infrastructure, NOT product code).
// Sparse LDL^T of a symmetric quasi-definite matrix: greedy minimum-degree ordering (stand-in for the AMD
// ordering Clarabel requests) + left-looking numeric factorisation with QDLDL-style dynamic regularisation
// (|D_ii| below eps is replaced by +-delta with the sign the quasi-definite structure prescribes).
// QDLDL is the direct solver behind Clarabel's default KKT path (third-party, absent from /root/reference;
// reference call site: /root/reference/mpc/qp/clarabel_interface.cpp:68-75).
#pragma once
#include <algorithm>
#include <cstdint>
#include <cmath>
#include <stdexcept>
#include <vector>

namespace orc {

struct SymTriplet { int r, c; double v; };   // one entry of the LOWER triangle (r >= c), original numbering

class SparseLDL {
public:
    // pattern: lower-triangle entries (values ignored).  Computes ordering + symbolic factor.
    void Analyze(int n, const std::vector<SymTriplet>& lower) {
        n_ = n;
        const int W = (n + 63) / 64;
        std::vector<uint64_t> adj((size_t)n * W, 0);
        auto setb = [&](int i, int j) { adj[(size_t)i * W + (j >> 6)] |= (1ull << (j & 63)); };
        for (auto& t : lower)
            if (t.r != t.c) { setb(t.r, t.c); setb(t.c, t.r); }
        std::vector<int> deg(n);
        auto popc = [&](int i) {
            int c = 0;
            for (int w = 0; w < W; w++) c += __builtin_popcountll(adj[(size_t)i * W + w]);
            return c;
        };
        for (int i = 0; i < n; i++) deg[i] = popc(i);
        std::vector<char> done(n, 0);
        perm_.assign(n, 0);      // perm_[pos] = original index
        iperm_.assign(n, 0);
        std::vector<std::vector<int>> colpat(n);   // by position, original indices of remaining neighbours
        std::vector<int> nb;
        for (int pos = 0; pos < n; pos++) {
            int p = -1, best = 1 << 30;
            for (int i = 0; i < n; i++)
                if (!done[i] && deg[i] < best) { best = deg[i]; p = i; }
            done[p] = 1;
            perm_[pos] = p;
            iperm_[p] = pos;
            nb.clear();
            const uint64_t* ap = &adj[(size_t)p * W];
            for (int w = 0; w < W; w++) {
                uint64_t x = ap[w];
                while (x) { int b = __builtin_ctzll(x); nb.push_back(w * 64 + b); x &= x - 1; }
            }
            colpat[pos] = nb;
            for (int u : nb) {
                uint64_t* au = &adj[(size_t)u * W];
                for (int w = 0; w < W; w++) au[w] |= ap[w];
                au[p >> 6] &= ~(1ull << (p & 63));
                au[u >> 6] &= ~(1ull << (u & 63));
                deg[u] = popc(u);
            }
        }
        // L column structure in permuted numbering, sorted
        Lp_.assign(n + 1, 0);
        for (int j = 0; j < n; j++) Lp_[j + 1] = Lp_[j] + (int)colpat[j].size();
        Li_.resize(Lp_[n]);
        for (int j = 0; j < n; j++) {
            int k = Lp_[j];
            for (int u : colpat[j]) Li_[k++] = iperm_[u];
            std::sort(Li_.begin() + Lp_[j], Li_.begin() + Lp_[j + 1]);
        }
        Lx_.assign(Lp_[n], 0.0);
        D_.assign(n, 0.0);
        // row lists: for row j, every column k<j with L(j,k) != 0 and the offset of that entry in column k
        rowp_.assign(n + 1, 0);
        for (int k = 0; k < n; k++)
            for (int e = Lp_[k]; e < Lp_[k + 1]; e++) rowp_[Li_[e] + 1]++;
        for (int j = 0; j < n; j++) rowp_[j + 1] += rowp_[j];
        rowk_.resize(rowp_[n]);
        rowe_.resize(rowp_[n]);
        std::vector<int> fill(rowp_.begin(), rowp_.end() - 1);
        for (int k = 0; k < n; k++)
            for (int e = Lp_[k]; e < Lp_[k + 1]; e++) {
                const int j = Li_[e];
                rowk_[fill[j]] = k;
                rowe_[fill[j]] = e;
                fill[j]++;
            }
        work_.assign(n, 0.0);
    }

    // values: same lower-triangle triplets (duplicates summed).  sign[i] = +1/-1 expected sign of D for original
    // index i (quasi-definite blocks).  Returns number of dynamically regularised pivots.
    int Factor(const std::vector<SymTriplet>& lower, const std::vector<int>& sign, double dyn_eps, double dyn_delta) {
        const int n = n_;
        // permuted lower-triangular A by column
        std::vector<std::vector<std::pair<int, double>>>& cols = acols_;
        cols.assign(n, {});
        for (auto& t : lower) {
            int i = iperm_[t.r], j = iperm_[t.c];
            if (i < j) std::swap(i, j);
            cols[j].push_back({i, t.v});
        }
        int nreg = 0;
        for (int j = 0; j < n; j++) {
            for (int e = Lp_[j]; e < Lp_[j + 1]; e++) work_[Li_[e]] = 0.0;
            work_[j] = 0.0;
            for (auto& pr : cols[j]) work_[pr.first] += pr.second;
            for (int r = rowp_[j]; r < rowp_[j + 1]; r++) {
                const int k = rowk_[r];
                const int e0 = rowe_[r];
                const double f = Lx_[e0] * D_[k];
                for (int e = e0; e < Lp_[k + 1]; e++) work_[Li_[e]] -= Lx_[e] * f;
            }
            double d = work_[j];
            const int sg = sign[perm_[j]];
            if (d * sg <= dyn_eps) { d = sg * dyn_delta; nreg++; }
            D_[j] = d;
            const double dinv = 1.0 / d;
            for (int e = Lp_[j]; e < Lp_[j + 1]; e++) Lx_[e] = work_[Li_[e]] * dinv;
        }
        return nreg;
    }

    // x <- K^-1 x  (original numbering)
    void Solve(std::vector<double>& x) const {
        const int n = n_;
        std::vector<double> y(n);
        for (int i = 0; i < n; i++) y[i] = x[perm_[i]];
        for (int j = 0; j < n; j++) {
            const double yj = y[j];
            for (int e = Lp_[j]; e < Lp_[j + 1]; e++) y[Li_[e]] -= Lx_[e] * yj;
        }
        for (int j = 0; j < n; j++) y[j] /= D_[j];
        for (int j = n - 1; j >= 0; j--) {
            double acc = y[j];
            for (int e = Lp_[j]; e < Lp_[j + 1]; e++) acc -= Lx_[e] * y[Li_[e]];
            y[j] = acc;
        }
        for (int i = 0; i < n; i++) x[perm_[i]] = y[i];
    }
    int nnzL() const { return Lp_.empty() ? 0 : Lp_[n_]; }

private:
    int n_ = 0;
    std::vector<int> perm_, iperm_, Lp_, Li_, rowp_, rowk_, rowe_;
    std::vector<double> Lx_, D_, work_;
    std::vector<std::vector<std::pair<int, double>>> acols_;
};

}  // namespace orc
