// piqp_amd/csrc/sparse_symbolic.cpp -- see sparse_symbolic.hpp
#include "sparse_symbolic.hpp"

#include <algorithm>
#include <cmath>
#include <stdexcept>

namespace pq {
namespace sparse {

// ------------------------------------------------------------------------------------------------
// Approximate minimum degree ordering on the pattern of A + A' (A upper triangular CSC, diagonal
// always included).  Published algorithm: Amestoy, Davis, Duff, "An approximate minimum degree ordering
// algorithm", SIAM J. Matrix Anal. Appl. 17(4) 1996, in the quotient-graph formulation of Davis,
// "Direct Methods for Sparse Linear Systems" (SIAM 2006, ch. 7) -- the same algorithm Eigen::AMDOrdering
// (called by the reference, sparse/ordering.hpp:72-74) implements.  perm[new] = old.
namespace {
inline int flip(int i) { return -i - 2; }

int wclear(int mark, int lemax, int* w, int n)
{
    if (mark < 2 || mark + lemax < 0) {
        for (int k = 0; k < n; ++k) if (w[k] != 0) w[k] = 1;
        mark = 2;
    }
    return mark;
}

int tree_dfs(int j, int k, int* head, const int* next, int* post, int* stack)
{
    int top = 0;
    stack[0] = j;
    while (top >= 0) {
        const int p = stack[top];
        const int i = head[p];
        if (i == -1) { --top; post[k++] = p; }
        else { head[p] = next[i]; stack[++top] = i; }
    }
    return k;
}
}  // namespace

void amd_order(int n, const int* Ap, const int* Ai, int* perm)
{
    if (n <= 0) return;
    IVec cnt(n + 1, 0);
    for (int j = 0; j < n; ++j) {
        for (int p = Ap[j]; p < Ap[j + 1]; ++p) { const int i = Ai[p]; if (i != j) { cnt[i]++; cnt[j]++; } }
        cnt[j]++;
    }
    long long cnz_ll = 0;
    for (int j = 0; j < n; ++j) cnz_ll += cnt[j];
    int cnz = (int)cnz_ll;
    const int nzmax = cnz + cnz / 5 + 2 * n;
    IVec Cp(n + 1, 0), Ci((size_t)nzmax + 1, 0);
    for (int j = 0; j < n; ++j) Cp[j + 1] = Cp[j] + cnt[j];
    IVec nxt(Cp.begin(), Cp.begin() + n);
    for (int j = 0; j < n; ++j) for (int p = Ap[j]; p < Ap[j + 1]; ++p) { const int i = Ai[p]; if (i < j) Ci[nxt[j]++] = i; }
    for (int j = 0; j < n; ++j) Ci[nxt[j]++] = j;
    for (int c = 0; c < n; ++c) for (int p = Ap[c]; p < Ap[c + 1]; ++p) { const int i = Ai[p]; if (i < c) Ci[nxt[i]++] = c; }

    int dense = (int)(10.0 * std::sqrt((double)n));
    dense = std::max(16, dense);
    dense = std::min(n - 2, dense);

    IVec len(n + 1), nv(n + 1, 1), next(n + 1, -1), head(n + 1, -1), elen(n + 1, 0), degree(n + 1), w(n + 1, 1), hhead(n + 1, -1), last(n + 1, -1), P(n + 1);
    for (int k = 0; k < n; ++k) len[k] = Cp[k + 1] - Cp[k];
    len[n] = 0;
    for (int i = 0; i <= n; ++i) degree[i] = len[i];
    int lemax = 0, nel = 0;
    int mark = wclear(0, 0, w.data(), n);
    for (int i = 0; i < n; ++i) {
        bool has_diag = false;
        for (int p = Cp[i]; p < Cp[i + 1]; ++p) if (Ci[p] == i) { has_diag = true; break; }
        const int d = degree[i];
        if (d == 1 && has_diag) { elen[i] = -2; nel++; Cp[i] = -1; w[i] = 0; }
        else if (d > dense || !has_diag) { nv[i] = 0; elen[i] = -1; nel++; Cp[i] = flip(n); nv[n]++; }
        else { if (head[d] != -1) last[head[d]] = i; next[i] = head[d]; head[d] = i; }
    }
    elen[n] = -2; Cp[n] = -1; w[n] = 0;

    int mindeg = 0;
    while (nel < n) {
        int k = -1;
        for (; mindeg < n && (k = head[mindeg]) == -1; ++mindeg) {}
        if (next[k] != -1) last[next[k]] = -1;
        head[mindeg] = next[k];
        const int elenk = elen[k];
        int nvk = nv[k];
        nel += nvk;
        if (elenk > 0 && cnz + mindeg >= nzmax) {  // compress the quotient graph storage
            for (int j = 0; j < n; ++j) { const int p = Cp[j]; if (p >= 0) { Cp[j] = Ci[p]; Ci[p] = flip(j); } }
            int q = 0;
            for (int p = 0; p < cnz;) {
                const int j = flip(Ci[p++]);
                if (j >= 0) {
                    Ci[q] = Cp[j];
                    Cp[j] = q++;
                    for (int k3 = 0; k3 < len[j] - 1; ++k3) Ci[q++] = Ci[p++];
                }
            }
            cnz = q;
        }
        int dk = 0;
        nv[k] = -nvk;
        int p = Cp[k];
        const int pk1 = (elenk == 0) ? p : cnz;
        int pk2 = pk1;
        for (int k1 = 1; k1 <= elenk + 1; ++k1) {
            int e, pj, ln;
            if (k1 > elenk) { e = k; pj = p; ln = len[k] - elenk; }
            else { e = Ci[p++]; pj = Cp[e]; ln = len[e]; }
            for (int k2 = 1; k2 <= ln; ++k2) {
                const int i = Ci[pj++];
                const int nvi = nv[i];
                if (nvi <= 0) continue;
                dk += nvi;
                nv[i] = -nvi;
                Ci[pk2++] = i;
                if (next[i] != -1) last[next[i]] = last[i];
                if (last[i] != -1) next[last[i]] = next[i]; else head[degree[i]] = next[i];
            }
            if (e != k) { Cp[e] = flip(k); w[e] = 0; }
        }
        if (elenk != 0) cnz = pk2;
        degree[k] = dk;
        Cp[k] = pk1;
        len[k] = pk2 - pk1;
        elen[k] = -2;

        mark = wclear(mark, lemax, w.data(), n);
        for (int pk = pk1; pk < pk2; ++pk) {
            const int i = Ci[pk];
            const int eln = elen[i];
            if (eln <= 0) continue;
            const int nvi = -nv[i];
            const int wnvi = mark - nvi;
            for (p = Cp[i]; p <= Cp[i] + eln - 1; ++p) {
                const int e = Ci[p];
                if (w[e] >= mark) w[e] -= nvi;
                else if (w[e] != 0) w[e] = degree[e] + wnvi;
            }
        }
        for (int pk = pk1; pk < pk2; ++pk) {
            const int i = Ci[pk];
            const int p1 = Cp[i];
            const int p2 = p1 + elen[i] - 1;
            int pn = p1;
            int h = 0, d = 0;
            for (p = p1; p <= p2; ++p) {
                const int e = Ci[p];
                if (w[e] != 0) {
                    const int dext = w[e] - mark;
                    if (dext > 0) { d += dext; Ci[pn++] = e; h += e; }
                    else { Cp[e] = flip(k); w[e] = 0; }
                }
            }
            elen[i] = pn - p1 + 1;
            const int p3 = pn;
            const int p4 = p1 + len[i];
            for (p = p2 + 1; p < p4; ++p) {
                const int j = Ci[p];
                const int nvj = nv[j];
                if (nvj <= 0) continue;
                d += nvj;
                Ci[pn++] = j;
                h += j;
            }
            if (d == 0) {
                Cp[i] = flip(k);
                const int nvi = -nv[i];
                dk -= nvi; nvk += nvi; nel += nvi;
                nv[i] = 0;
                elen[i] = -1;
            } else {
                degree[i] = std::min(degree[i], d);
                Ci[pn] = Ci[p3];
                Ci[p3] = Ci[p1];
                Ci[p1] = k;
                len[i] = pn - p1 + 1;
                h = (h < 0) ? -h : h;
                h %= n;
                next[i] = hhead[h];
                hhead[h] = i;
                last[i] = h;
            }
        }
        degree[k] = dk;
        lemax = std::max(lemax, dk);
        mark = wclear(mark + lemax, lemax, w.data(), n);

        for (int pk = pk1; pk < pk2; ++pk) {
            int i = Ci[pk];
            if (nv[i] >= 0) continue;
            const int h = last[i];
            i = hhead[h];
            hhead[h] = -1;
            for (; i != -1 && next[i] != -1; i = next[i], ++mark) {
                const int ln = len[i];
                const int eln = elen[i];
                for (p = Cp[i] + 1; p <= Cp[i] + ln - 1; ++p) w[Ci[p]] = mark;
                int jlast = i;
                for (int j = next[i]; j != -1;) {
                    bool ok = (len[j] == ln) && (elen[j] == eln);
                    for (p = Cp[j] + 1; ok && p <= Cp[j] + ln - 1; ++p) if (w[Ci[p]] != mark) ok = false;
                    if (ok) {
                        Cp[j] = flip(i);
                        nv[i] += nv[j];
                        nv[j] = 0;
                        elen[j] = -1;
                        j = next[j];
                        next[jlast] = j;
                    } else { jlast = j; j = next[j]; }
                }
            }
        }
        int pw = pk1;
        for (int pk = pk1; pk < pk2; ++pk) {
            const int i = Ci[pk];
            const int nvi = -nv[i];
            if (nvi <= 0) continue;
            nv[i] = nvi;
            int d = degree[i] + dk - nvi;
            d = std::min(d, n - nel - nvi);
            if (head[d] != -1) last[head[d]] = i;
            next[i] = head[d];
            last[i] = -1;
            head[d] = i;
            mindeg = std::min(mindeg, d);
            degree[i] = d;
            Ci[pw++] = i;
        }
        nv[k] = nvk;
        if ((len[k] = pw - pk1) == 0) { Cp[k] = -1; w[k] = 0; }
        if (elenk != 0) cnz = pw;
    }
    for (int i = 0; i < n; ++i) Cp[i] = flip(Cp[i]);
    for (int j = 0; j <= n; ++j) head[j] = -1;
    for (int j = n; j >= 0; --j) { if (nv[j] > 0) continue; next[j] = head[Cp[j]]; head[Cp[j]] = j; }
    for (int e = n; e >= 0; --e) { if (nv[e] <= 0) continue; if (Cp[e] != -1) { next[e] = head[Cp[e]]; head[Cp[e]] = e; } }
    int k = 0;
    for (int i = 0; i <= n; ++i) if (Cp[i] == -1) k = tree_dfs(i, k, head.data(), next.data(), P.data(), w.data());
    for (int i = 0; i < n; ++i) perm[i] = P[i];
}

// ------------------------------------------------------------------------------------------------
// C = upper(P A P') with sorted rows; Ai_to_Ci maps value positions (sparse/utils.hpp:32-128)
static void permute_sym_upper(int n, const IVec& Ap, const IVec& Ai, const int* perm_inv, IVec& Cp, IVec& Ci, IVec& Ai_to_Ci)
{
    const int nnz = Ap[n];
    IVec w(n, 0);
    for (int j = 0; j < n; ++j) {
        const int j2 = perm_inv[j];
        for (int p = Ap[j]; p < Ap[j + 1]; ++p) { const int i = Ai[p]; if (i > j) continue; const int i2 = perm_inv[i]; w[std::min(i2, j2)]++; }
    }
    IVec CTp(n + 1), CTi(nnz), CT_to_A(nnz);
    int sum = 0;
    for (int i = 0; i < n; ++i) { CTp[i] = sum; sum += w[i]; w[i] = CTp[i]; }
    CTp[n] = sum;
    for (int j = 0; j < n; ++j) {
        const int j2 = perm_inv[j];
        for (int k = Ap[j]; k < Ap[j + 1]; ++k) {
            const int i = Ai[k];
            if (i > j) continue;
            const int i2 = perm_inv[i];
            const int q = w[std::min(i2, j2)]++;
            CTi[q] = std::max(i2, j2);
            CT_to_A[q] = k;
        }
    }
    Cp.assign(n + 1, 0); Ci.assign(nnz, 0); Ai_to_Ci.assign(nnz, 0);
    for (int j = 0; j < n; ++j) for (int p = CTp[j]; p < CTp[j + 1]; ++p) Cp[CTi[p]]++;
    sum = 0;
    for (int j = 0; j < n; ++j) { const int t = Cp[j]; Cp[j] = sum; w[j] = sum; sum += t; }
    Cp[n] = sum;
    for (int j = 0; j < n; ++j)
        for (int k = CTp[j]; k < CTp[j + 1]; ++k) { const int q = w[CTi[k]]++; Ci[q] = j; Ai_to_Ci[CT_to_A[k]] = q; }
}

// elimination tree of an upper-triangular CSC pattern (sparse/ldlt.hpp:61-83 computes the same parent array)
static void elimination_tree(int n, const IVec& Cp, const IVec& Ci, IVec& parent, IVec& colcount)
{
    parent.assign(n, -1);
    colcount.assign(n, 0);
    IVec flag(n);
    for (int k = 0; k < n; ++k) {
        flag[k] = k;
        for (int p = Cp[k]; p < Cp[k + 1]; ++p) {
            int i = Ci[p];
            for (; flag[i] != k; i = parent[i]) {
                if (parent[i] == -1) parent[i] = k;
                colcount[i]++;
                flag[i] = k;
            }
        }
    }
}

static void postorder(int n, const IVec& parent, IVec& post)
{
    IVec head(n, -1), next(n, -1), stack(n);
    for (int j = n - 1; j >= 0; --j) { if (parent[j] == -1) continue; next[j] = head[parent[j]]; head[parent[j]] = j; }
    post.assign(n, 0);
    int k = 0;
    for (int j = 0; j < n; ++j) {
        if (parent[j] != -1) continue;
        int top = 0;
        stack[0] = j;
        while (top >= 0) {
            const int p = stack[top];
            const int i = head[p];
            if (i == -1) { --top; post[k++] = p; }
            else { head[p] = next[i]; stack[++top] = i; }
        }
    }
}

void analyse_kkt_full(const pq_sparse_data* d, Symbolic& S)
{
    const int n = d->n, p = d->p, m = d->m, N = n + p + m;
    S.n = n; S.p = p; S.m = m; S.N = N;
    const int* Pp = d->P_colptr; const int* Pi = d->P_rowind; const double* Px = d->P_val;
    const int* Atp = d->AT_colptr; const int* Ati = d->AT_rowind; const double* Atx = d->AT_val;
    const int* Gtp = d->GT_colptr; const int* Gti = d->GT_rowind; const double* Gtx = d->GT_val;
    const int nzP = Pp[n], nzA = p ? Atp[p] : 0, nzG = m ? Gtp[m] : 0;

    // ---- K = [P+rho, AT, GT; ., -delta, .; ., ., -(W+delta)] upper, diagonal last in every column (kkt_full.hpp:39-170)
    S.Kp.assign(N + 1, 0);
    S.P_utri_to_Ki.assign(nzP, 0); S.AT_to_Ki.assign(nzA, 0); S.GT_to_Ki.assign(nzG, 0);
    int nz = 0, jk = 0;
    for (int j = 0; j < n; ++j) {
        int c = Pp[j + 1] - Pp[j];
        if (c == 0 || Pi[Pp[j + 1] - 1] != j) c += 1;
        nz += c; S.Kp[++jk] = nz;
    }
    for (int j = 0; j < p; ++j) { nz += Atp[j + 1] - Atp[j] + 1; S.Kp[++jk] = nz; }
    for (int j = 0; j < m; ++j) { nz += Gtp[j + 1] - Gtp[j] + 1; S.Kp[++jk] = nz; }
    S.Ki.assign(nz, 0); S.Kx.assign(nz, 0.0);
    jk = 0;
    for (int j = 0; j < n; ++j, ++jk) {
        const int kk = S.Kp[jk], c = Pp[j + 1] - Pp[j];
        for (int q = 0; q < c; ++q) { S.Ki[kk + q] = Pi[Pp[j] + q]; S.Kx[kk + q] = Px[Pp[j] + q]; S.P_utri_to_Ki[Pp[j] + q] = kk + q; }
        const int kc = S.Kp[jk + 1] - kk;
        if (kc > c) { S.Ki[kk + kc - 1] = jk; S.Kx[kk + kc - 1] = 0.0; }
    }
    for (int j = 0; j < p; ++j, ++jk) {
        const int kk = S.Kp[jk], c = Atp[j + 1] - Atp[j];
        for (int q = 0; q < c; ++q) { S.Ki[kk + q] = Ati[Atp[j] + q]; S.Kx[kk + q] = Atx[Atp[j] + q]; S.AT_to_Ki[Atp[j] + q] = kk + q; }
        S.Ki[kk + c] = jk; S.Kx[kk + c] = 0.0;
    }
    for (int j = 0; j < m; ++j, ++jk) {
        const int kk = S.Kp[jk], c = Gtp[j + 1] - Gtp[j];
        for (int q = 0; q < c; ++q) { S.Ki[kk + q] = Gti[Gtp[j] + q]; S.Kx[kk + q] = Gtx[Gtp[j] + q]; S.GT_to_Ki[Gtp[j] + q] = kk + q; }
        S.Ki[kk + c] = jk; S.Kx[kk + c] = 0.0;
    }

    // ---- fill-reducing ordering, then postorder of the elimination tree (keeps the fill, makes subtrees contiguous)
    IVec perm0(N), pinv0(N);
    amd_order(N, S.Kp.data(), S.Ki.data(), perm0.data());
    for (int i = 0; i < N; ++i) pinv0[perm0[i]] = i;
    {
        IVec Cp0, Ci0, map0, parent0, cc0, post;
        permute_sym_upper(N, S.Kp, S.Ki, pinv0.data(), Cp0, Ci0, map0);
        elimination_tree(N, Cp0, Ci0, parent0, cc0);
        postorder(N, parent0, post);
        S.P.assign(N, 0); S.P_inv.assign(N, 0);
        for (int k = 0; k < N; ++k) S.P[k] = perm0[post[k]];
        for (int k = 0; k < N; ++k) S.P_inv[S.P[k]] = k;
    }
    permute_sym_upper(N, S.Kp, S.Ki, S.P_inv.data(), S.Cp, S.Ci, S.PKi);
    S.diag_pos.assign(N, 0);
    for (int col = 0; col < N; ++col) S.diag_pos[col] = S.Cp[S.P_inv[col] + 1] - 1;

    // ---- column structure of L (row lists, increasing)
    IVec cc;
    elimination_tree(N, S.Cp, S.Ci, S.etree, cc);
    std::vector<long long> Lp(N + 1, 0);
    for (int j = 0; j < N; ++j) Lp[j + 1] = Lp[j] + cc[j];
    S.nnzL = Lp[N];
    IVec Li((size_t)S.nnzL), fill(N, 0), flag(N);
    for (int k = 0; k < N; ++k) {
        flag[k] = k;
        for (int q = S.Cp[k]; q < S.Cp[k + 1]; ++q) {
            int i = S.Ci[q];
            for (; flag[i] != k; i = S.etree[i]) { Li[Lp[i] + fill[i]++] = k; flag[i] = k; }
        }
    }
    S.flops = 0.0;
    for (int j = 0; j < N; ++j) S.flops += (double)cc[j] * cc[j] + 3.0 * cc[j];

    // ---- supernodes: j+1 joins j when parent(j) = j+1 and struct(L_j) \ {j+1} = struct(L_{j+1})
    S.sn_of_col.assign(N, 0);
    S.sn_first.clear();
    for (int j = 0; j < N; ++j) {
        const bool joins = j > 0 && S.etree[j - 1] == j && cc[j - 1] == cc[j] + 1;
        if (!joins) S.sn_first.push_back(j);
        S.sn_of_col[j] = (int)S.sn_first.size() - 1;
    }
    S.nsuper = (int)S.sn_first.size();
    S.sn_first.push_back(N);
    const int ns = S.nsuper;
    S.sn_parent.assign(ns, -1);
    S.front_rows_ptr.assign(ns + 1, 0);
    S.front_off.assign(ns + 1, 0);
    S.max_front = 0;
    for (int s = 0; s < ns; ++s) {
        const int last = S.sn_first[s + 1] - 1, w = S.sn_first[s + 1] - S.sn_first[s];
        const int f = w + cc[last];
        S.front_rows_ptr[s + 1] = S.front_rows_ptr[s] + f;
        S.front_off[s + 1] = S.front_off[s] + (long long)f * f;
        S.max_front = std::max(S.max_front, f);
        if (S.etree[last] >= 0) S.sn_parent[s] = S.sn_of_col[S.etree[last]];
    }
    S.front_doubles = S.front_off[ns];
    S.front_rows.assign(S.front_rows_ptr[ns], 0);
    for (int s = 0; s < ns; ++s) {
        const int first = S.sn_first[s], last = S.sn_first[s + 1] - 1, w = last - first + 1;
        int* r = S.front_rows.data() + S.front_rows_ptr[s];
        for (int i = 0; i < w; ++i) r[i] = first + i;
        for (int i = 0; i < cc[last]; ++i) r[w + i] = Li[Lp[last] + i];
    }

    // ---- children lists and levels
    S.child_ptr.assign(ns + 1, 0);
    for (int s = 0; s < ns; ++s) if (S.sn_parent[s] >= 0) S.child_ptr[S.sn_parent[s] + 1]++;
    for (int s = 0; s < ns; ++s) S.child_ptr[s + 1] += S.child_ptr[s];
    S.child.assign(S.child_ptr[ns], 0);
    {
        IVec nx(S.child_ptr.begin(), S.child_ptr.end() - 1);
        for (int s = 0; s < ns; ++s) if (S.sn_parent[s] >= 0) S.child[nx[S.sn_parent[s]]++] = s;
    }
    IVec level(ns, 0);
    int maxlev = 0;
    for (int s = 0; s < ns; ++s) {  // children precede parents (postorder)
        const int ps = S.sn_parent[s];
        if (ps >= 0) level[ps] = std::max(level[ps], level[s] + 1);
        maxlev = std::max(maxlev, level[s]);
    }
    S.nlevels = ns ? maxlev + 1 : 0;
    S.level_ptr.assign(S.nlevels + 1, 0);
    for (int s = 0; s < ns; ++s) S.level_ptr[level[s] + 1]++;
    for (int l = 0; l < S.nlevels; ++l) S.level_ptr[l + 1] += S.level_ptr[l];
    S.level_sn.assign(ns, 0);
    {
        IVec nx(S.level_ptr.begin(), S.level_ptr.end() - 1);
        for (int s = 0; s < ns; ++s) S.level_sn[nx[level[s]]++] = s;
    }

    // ---- assembly map: upper entry (i, k), i <= k  ==  lower entry (k, i) of the front that owns column i
    S.a_dst.assign(S.Cp[N], 0);
    for (int k = 0; k < N; ++k) {
        for (int q = S.Cp[k]; q < S.Cp[k + 1]; ++q) {
            const int i = S.Ci[q];
            const int s = S.sn_of_col[i];
            const int first = S.sn_first[s], last = S.sn_first[s + 1] - 1, w = last - first + 1;
            const int f = S.front_rows_ptr[s + 1] - S.front_rows_ptr[s];
            int lrow;
            if (k <= last) lrow = k - first;
            else {
                const int* r = S.front_rows.data() + S.front_rows_ptr[s] + w;
                const int* e = r + (f - w);
                const int* it = std::lower_bound(r, e, k);
                if (it == e || *it != k) throw std::runtime_error("symbolic: entry outside the front structure");
                lrow = w + (int)(it - r);
            }
            S.a_dst[q] = S.front_off[s] + lrow + (long long)(i - first) * f;
        }
    }

    // ---- extend-add maps
    S.rel_ptr.assign(ns + 1, 0);
    for (int c = 0; c < ns; ++c) {
        const int w = S.sn_first[c + 1] - S.sn_first[c];
        const int f = S.front_rows_ptr[c + 1] - S.front_rows_ptr[c];
        S.rel_ptr[c + 1] = S.rel_ptr[c] + (S.sn_parent[c] >= 0 ? f - w : 0);
    }
    S.rel.assign(S.rel_ptr[ns], 0);
    for (int c = 0; c < ns; ++c) {
        const int ps = S.sn_parent[c];
        if (ps < 0) continue;
        const int w = S.sn_first[c + 1] - S.sn_first[c];
        const int f = S.front_rows_ptr[c + 1] - S.front_rows_ptr[c];
        const int* rc = S.front_rows.data() + S.front_rows_ptr[c] + w;
        const int* rp = S.front_rows.data() + S.front_rows_ptr[ps];
        const int fp = S.front_rows_ptr[ps + 1] - S.front_rows_ptr[ps];
        int pos = 0;
        for (int i = 0; i < f - w; ++i) {
            while (pos < fp && rp[pos] < rc[i]) ++pos;
            if (pos >= fp || rp[pos] != rc[i]) throw std::runtime_error("symbolic: child update row missing in parent front");
            S.rel[S.rel_ptr[c] + i] = pos;
        }
    }
}

}  // namespace sparse
}  // namespace pq
