// piqp_amd/csrc/sparse_symbolic.cpp -- see sparse_symbolic.hpp
#include "sparse_symbolic.hpp"

#include <algorithm>
#include <cstdio>
#include <cmath>
#include <cstdlib>
#include <exception>
#include <stdexcept>
#include <string>
#include <thread>

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
// Nested dissection by BFS level structures (George & Liu, "Computer Solution of Large Sparse Positive Definite
// Systems", 1981, ch. 8: automatic nested dissection): a connected part is split by the middle level of the rooted
// level structure of a pseudo-peripheral node; parts below `leaf` nodes are ordered by AMD on their induced subgraph,
// separators are numbered after both halves.  Not in the reference: its up-looking LDLt is serial, so it only cares
// about fill (AMD).  On the device the numeric phase runs one assembly-tree LEVEL at a time, so the depth of the
// tree is the length of the critical path; for chain-like KKT graphs (banded / staged problems) AMD produces trees
// thousands of levels deep while dissection gives O(log N) separator levels on top of shallow leaves.
void nd_order(int n, const int* Ap, const int* Ai, int* perm, int leaf)
{
    if (n <= 0) return;
    // symmetric adjacency without the diagonal
    IVec xadj(n + 1, 0);
    for (int j = 0; j < n; ++j) for (int p = Ap[j]; p < Ap[j + 1]; ++p) { const int i = Ai[p]; if (i != j) { xadj[i + 1]++; xadj[j + 1]++; } }
    for (int j = 0; j < n; ++j) xadj[j + 1] += xadj[j];
    IVec adj(xadj[n]), nx(xadj.begin(), xadj.end() - 1);
    for (int j = 0; j < n; ++j) for (int p = Ap[j]; p < Ap[j + 1]; ++p) { const int i = Ai[p]; if (i != j) { adj[nx[i]++] = j; adj[nx[j]++] = i; } }

    IVec part(n, 0);      // current part id of every node (-1 = already numbered)
    IVec level(n, -1), queue(n), local(n, -1);
    int next_part = 1, pos = 0;
    struct Item { int part; std::vector<int> nodes; bool is_sep; };
    std::vector<Item> stack;
    // "dense" nodes (global variables / coupling constraints: degree far above the average) would put every other node within two
    // hops of each other and flatten every level structure; they are taken out of the graph and numbered last, like the
    // dense-row treatment of AMD and like the arrow block of the multistage backend
    std::vector<int> dense_nodes;
    {
        const double avg = n ? (double)xadj[n] / n : 0.0;
        const int thresh = std::max(32, (int)(10.0 * avg));
        Item all; all.part = 0; all.is_sep = false;
        for (int i = 0; i < n; ++i) {
            if (xadj[i + 1] - xadj[i] > thresh) { dense_nodes.push_back(i); part[i] = -3; }
            else all.nodes.push_back(i);
        }
        if (!all.nodes.empty()) stack.push_back(std::move(all));
    }
    // rooted level structure of `root` inside part `pid`; returns number of levels, nodes in BFS order in queue[0..cnt)
    auto bfs = [&](int root, int pid, int& cnt) {
        int head = 0;
        cnt = 0;
        queue[cnt++] = root; level[root] = 0;
        int maxl = 0;
        while (head < cnt) {
            const int v = queue[head++];
            for (int q = xadj[v]; q < xadj[v + 1]; ++q) {
                const int wn = adj[q];
                if (part[wn] == pid && level[wn] < 0) { level[wn] = level[v] + 1; maxl = level[wn]; queue[cnt++] = wn; }
            }
        }
        return maxl + 1;
    };
    auto number_leaf = [&](const std::vector<int>& nodes) {
        const int k = (int)nodes.size();
        if (k <= 2) { for (int v : nodes) perm[pos++] = v; return; }
        for (int i = 0; i < k; ++i) local[nodes[i]] = i;
        IVec lp(k + 1, 0), li;
        for (int i = 0; i < k; ++i) {  // upper pattern of the induced subgraph, diagonal included
            const int v = nodes[i];
            for (int q = xadj[v]; q < xadj[v + 1]; ++q) { const int l = local[adj[q]]; if (l >= 0 && l < i) li.push_back(l); }
            li.push_back(i);
            lp[i + 1] = (int)li.size();
        }
        IVec lperm(k);
        amd_order(k, lp.data(), li.data(), lperm.data());
        for (int i = 0; i < k; ++i) perm[pos++] = nodes[lperm[i]];
        for (int i = 0; i < k; ++i) local[nodes[i]] = -1;
    };
    // the stack is processed so that, for every dissection, both halves are numbered before their separator
    while (!stack.empty()) {
        Item it = std::move(stack.back());
        stack.pop_back();
        if (it.is_sep) { for (int v : it.nodes) { perm[pos++] = v; part[v] = -1; } continue; }
        if ((int)it.nodes.size() <= leaf) { number_leaf(it.nodes); for (int v : it.nodes) part[v] = -1; continue; }
        const int pid = it.part;
        // connected component of the first node
        int cnt = 0;
        bfs(it.nodes[0], pid, cnt);
        if (cnt < (int)it.nodes.size()) {
            // disconnected: ALL components of the part in one sweep, numbered in the order of their first node (splitting off one component and
            // revisiting "the rest" is quadratic in the number of components: the 93 261 isolated variables that remain of MM BOYD1 once its 18
            // dense rows are set aside took 10 s)
            std::vector<Item> comps;
            auto take = [&](int count) {
                Item comp; comp.part = next_part++; comp.is_sep = false;
                comp.nodes.assign(queue.begin(), queue.begin() + count);
                for (int q = 0; q < count; ++q) { part[queue[q]] = comp.part; level[queue[q]] = -1; }
                comps.push_back(std::move(comp));
            };
            take(cnt);
            for (int v : it.nodes) {
                if (part[v] != pid) continue;
                int c2 = 0;
                bfs(v, pid, c2);
                take(c2);
            }
            for (size_t q = comps.size(); q-- > 0;) stack.push_back(std::move(comps[q]));  // popped in the order found
            continue;
        }
        // pseudo-peripheral root: restart from a node of the last level while the structure gets deeper
        int nlev = 0, root = queue[cnt - 1];
        for (int sweep = 0; sweep < 4; ++sweep) {
            for (int q = 0; q < cnt; ++q) level[queue[q]] = -1;
            const int nl = bfs(root, pid, cnt);
            if (nl <= nlev) { nlev = nl; break; }
            nlev = nl;
            root = queue[cnt - 1];
        }
        if (nlev < 3) {  // no interior level to cut at
            for (int q = 0; q < cnt; ++q) level[queue[q]] = -1;
            number_leaf(it.nodes);
            for (int v : it.nodes) part[v] = -1;
            continue;
        }
        // cut level: the smallest level within the middle third (by node count)
        IVec lsize(nlev, 0);
        for (int q = 0; q < cnt; ++q) lsize[level[queue[q]]]++;
        int cum = 0, lo = 1, hi = nlev - 2;
        for (int l = 0; l < nlev; ++l) { cum += lsize[l]; if (3 * cum < cnt) lo = std::max(lo, l + 1); if (3 * cum <= 2 * cnt) hi = std::min(nlev - 2, std::max(l, 1)); }
        if (hi < lo) hi = lo = std::min(std::max(1, (lo + hi) / 2), nlev - 2);
        int cut = lo;
        for (int l = lo; l <= hi; ++l) if (lsize[l] < lsize[cut]) cut = l;
        Item A, B, Sp;
        A.part = next_part++; B.part = next_part++; Sp.part = -1;
        A.is_sep = B.is_sep = false; Sp.is_sep = true;
        for (int q = 0; q < cnt; ++q) {
            const int v = queue[q];
            if (level[v] < cut) { A.nodes.push_back(v); part[v] = A.part; }
            else if (level[v] > cut) { B.nodes.push_back(v); part[v] = B.part; }
            else { Sp.nodes.push_back(v); part[v] = -2; }
            level[v] = -1;
        }
        stack.push_back(std::move(Sp));  // popped last
        stack.push_back(std::move(B));
        stack.push_back(std::move(A));
    }
    for (int v : dense_nodes) perm[pos++] = v;
}

// ------------------------------------------------------------------------------------------------
// C = upper(P A P') with sorted rows; Ai_to_Ci maps value positions (sparse/utils.hpp:32-128)
void permute_sym_upper(int n, const IVec& Ap, const IVec& Ai, const int* perm_inv, IVec& Cp, IVec& Ci, IVec& Ai_to_Ci)
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

static void analyse_with_order(Symbolic& S, const IVec& perm0, const IVec* forced_first = nullptr);

// structural upper triangle of MT * MT^T (MT: n x k CSC) with, for every entry (i <= j), the list of value-index pairs
// (q_i, q_j) of the constraints k that contain both variables, constraints ascending -- the summation order of the
// reference's update_AT_A / update_GT_W_delta_inv_G (kkt_all_eliminated.hpp:184-223)
static void gram_structure(int n, int k, const int* MTp, const int* MTi, Symbolic::Gram& Gm)
{
    // row-oriented view: for variable j the (constraint, value index) pairs, constraints ascending
    IVec rp(n + 1, 0);
    const int nnz = k ? MTp[k] : 0;
    for (int q = 0; q < nnz; ++q) rp[MTi[q] + 1]++;
    for (int j = 0; j < n; ++j) rp[j + 1] += rp[j];
    {   // the term lists hold one triple per pair of entries of a constraint row: sum_c nnz_c (nnz_c + 1) / 2 of them.  Constraint rows with thousands of
        // entries (an arrow of dense rows) make that billions -- the reference's A^T A would be a dense n x n matrix there as well; refuse instead of
        // exhausting the host's memory (round 4: a 300 x 20 000 dense equality block asked for 6e10 triples and took the machine down)
        double terms = 0.0;
        for (int c = 0; c < k; ++c) { const double r = MTp[c + 1] - MTp[c]; terms += 0.5 * r * (r + 1.0); }
        if (terms > 2.0e8) throw std::runtime_error("condensed KKT mode: the product pattern of the eliminated block needs more than 2e8 terms (constraint rows too dense); use sparse_ldlt (KKT_FULL)");
    }
    IVec rk(nnz), rq(nnz), nx(rp.begin(), rp.end() - 1);
    for (int c = 0; c < k; ++c) for (int q = MTp[c]; q < MTp[c + 1]; ++q) { const int t = nx[MTi[q]]++; rk[t] = c; rq[t] = q; }
    Gm.colptr.assign(n + 1, 0);
    Gm.rowind.clear(); Gm.ptr.assign(1, 0); Gm.q1.clear(); Gm.q2.clear(); Gm.k.clear();
    IVec head(n, -1);  // head[i] = entry index of (i, j) in the current column
    std::vector<std::vector<int>> tmp;  // per entry of the current column: (q1, q2, k) triples
    for (int j = 0; j < n; ++j) {
        IVec rows;
        for (int t = rp[j]; t < rp[j + 1]; ++t) {
            const int c = rk[t], qj = rq[t];
            for (int q = MTp[c]; q < MTp[c + 1]; ++q) {
                const int i = MTi[q];
                if (i > j) continue;
                if (head[i] < 0) { head[i] = (int)rows.size(); rows.push_back(i); tmp.emplace_back(); if ((int)tmp.size() < (int)rows.size()) tmp.resize(rows.size()); }
                auto& L = tmp[head[i]];
                L.push_back(q); L.push_back(qj); L.push_back(c);
            }
        }
        IVec order(rows.size());
        for (size_t a = 0; a < rows.size(); ++a) order[a] = (int)a;
        std::sort(order.begin(), order.end(), [&](int a, int b2) { return rows[a] < rows[b2]; });
        for (int a : order) {
            Gm.rowind.push_back(rows[a]);
            const auto& L = tmp[a];
            for (size_t t = 0; t < L.size(); t += 3) { Gm.q1.push_back(L[t]); Gm.q2.push_back(L[t + 1]); Gm.k.push_back(L[t + 2]); }
            Gm.ptr.push_back((int)Gm.q1.size());
        }
        for (int i : rows) head[i] = -1;
        for (size_t a = 0; a < rows.size(); ++a) tmp[a].clear();
        Gm.colptr[j + 1] = (int)Gm.rowind.size();
    }
}

void analyse_kkt_full(const pq_sparse_data* d, Symbolic& S) { analyse_kkt(d, 0, S); }
static void order_and_analyse(Symbolic& S);

// mode: KKTMode bits (kkt_fwd.hpp:15-21): 1 = equalities eliminated, 2 = inequalities eliminated
void analyse_kkt(const pq_sparse_data* d, int mode, Symbolic& S)
{
    analyse_kkt_pattern(d, mode, S);
    order_and_analyse(S);
}

// K of the mode and the value maps into it (create_kkt_matrix)
void analyse_kkt_pattern(const pq_sparse_data* d, int mode, Symbolic& S)
{
    const int n = d->n, p = d->p, m = d->m;
    const bool eq = (mode & 1) != 0, ineq = (mode & 2) != 0;
    const int N = n + (eq ? 0 : p) + (ineq ? 0 : m);
    S.n = n; S.p = p; S.m = m; S.N = N; S.mode = mode;
    static const int zero_ptr[1] = {0};
    const int* Pp = d->P_colptr; const int* Pi = d->P_rowind; const double* Px = d->P_val;
    const int* Atp = p ? d->AT_colptr : zero_ptr; const int* Ati = d->AT_rowind; const double* Atx = d->AT_val;
    const int* Gtp = m ? d->GT_colptr : zero_ptr; const int* Gti = d->GT_rowind; const double* Gtx = d->GT_val;
    const int nzP = Pp[n], nzA = p ? Atp[p] : 0, nzG = m ? Gtp[m] : 0;
    if (eq) gram_structure(n, p, Atp, Ati, S.gramA); else S.gramA = Symbolic::Gram();
    if (ineq) gram_structure(n, m, Gtp, Gti, S.gramG); else S.gramG = Symbolic::Gram();

    // ---- K upper, diagonal last in every column.  Top-left block: P_utri + I (+ A'A) (+ G'G) as a sorted merge
    // (kkt_{eq,ineq,all}_eliminated.hpp create_kkt_matrix; kkt_full.hpp:39-170 when nothing is eliminated); then one column
    // per kept constraint: [AT col; -delta] and [GT col; -z_reg]
    S.Kp.assign(N + 1, 0);
    S.Ki.clear(); S.Kx.clear();
    S.P_utri_to_Ki.assign(nzP, 0); S.AT_to_Ki.assign(eq ? 0 : nzA, 0); S.GT_to_Ki.assign(ineq ? 0 : nzG, 0);
    S.gramA_to_Ki.assign(eq ? S.gramA.rowind.size() : 0, 0); S.gramG_to_Ki.assign(ineq ? S.gramG.rowind.size() : 0, 0);
    for (int j = 0; j < n; ++j) {
        int a = Pp[j], ae = Pp[j + 1];
        int b = eq ? S.gramA.colptr[j] : 0, be = eq ? S.gramA.colptr[j + 1] : 0;
        int c = ineq ? S.gramG.colptr[j] : 0, ce = ineq ? S.gramG.colptr[j + 1] : 0;
        bool diag_done = false;
        for (;;) {
            int r = n + 1;
            if (a < ae) r = std::min(r, Pi[a]);
            if (b < be) r = std::min(r, S.gramA.rowind[b]);
            if (c < ce) r = std::min(r, S.gramG.rowind[c]);
            if (!diag_done) r = std::min(r, j);
            if (r > n) break;
            const int at = (int)S.Ki.size();
            double v = 0.0;
            if (a < ae && Pi[a] == r) { v = Px[a]; S.P_utri_to_Ki[a++] = at; }
            if (b < be && S.gramA.rowind[b] == r) S.gramA_to_Ki[b++] = at;
            if (c < ce && S.gramG.rowind[c] == r) S.gramG_to_Ki[c++] = at;
            if (r == j) diag_done = true;
            if (r > j) throw std::runtime_error("symbolic: P is not upper triangular");
            S.Ki.push_back(r); S.Kx.push_back(v);
        }
        S.Kp[j + 1] = (int)S.Ki.size();
    }
    int jk = n;
    if (!eq)
        for (int j = 0; j < p; ++j, ++jk) {
            for (int q = Atp[j]; q < Atp[j + 1]; ++q) { S.AT_to_Ki[q] = (int)S.Ki.size(); S.Ki.push_back(Ati[q]); S.Kx.push_back(Atx[q]); }
            S.Ki.push_back(jk); S.Kx.push_back(0.0);
            S.Kp[jk + 1] = (int)S.Ki.size();
        }
    if (!ineq)
        for (int j = 0; j < m; ++j, ++jk) {
            for (int q = Gtp[j]; q < Gtp[j + 1]; ++q) { S.GT_to_Ki[q] = (int)S.Ki.size(); S.Ki.push_back(Gti[q]); S.Kx.push_back(Gtx[q]); }
            S.Ki.push_back(jk); S.Kx.push_back(0.0);
            S.Kp[jk + 1] = (int)S.Ki.size();
        }

}

static void order_and_analyse(Symbolic& S)
{
    const int N = S.N;
    // ---- ordering: AMD (what the reference uses, sparse/ordering.hpp:72-74) or nested dissection, whichever gives the
    // cheaper device schedule (levels = dependent kernel launches; fill = HBM traffic and flops)
    const char* want = std::getenv("PIQP_AMD_ORDERING");
    const std::string ord = want ? want : "auto";
    IVec perm_amd(N);
    if (ord == "amd" || (ord == "auto" && N < 200)) {
        amd_order(N, S.Kp.data(), S.Ki.data(), perm_amd.data());
        analyse_with_order(S, perm_amd); S.ordering = "amd"; S.fill_perm = perm_amd; return;
    }
    IVec perm_nd(N);
    const int nd_leaf = 256;  // measured with the leaf amalgamation: 96 -> 256 is +8 % on C3 and +5 % on the n = 500k chain, 384 falls off a cliff
    Symbolic T = S;
    auto do_nd = [&] {
        nd_order(N, T.Kp.data(), T.Ki.data(), perm_nd.data(), nd_leaf);
        analyse_with_order(T, perm_nd);
        T.ordering = "nested dissection";
        T.fill_perm = perm_nd;
    };
    if (ord == "nd") { do_nd(); S = std::move(T); return; }
    // the two candidate analyses are independent host work (0.2 s each at N = 100 000): side by side on two threads (round 4)
    (void)debug_token("tree_profile");  // (the token list is parsed once, here, before the second thread exists)
    std::exception_ptr nd_err;
    std::thread nd_thread([&] { try { do_nd(); } catch (...) { nd_err = std::current_exception(); } });
    try {
        amd_order(N, S.Kp.data(), S.Ki.data(), perm_amd.data());
        analyse_with_order(S, perm_amd);
    } catch (...) { nd_thread.join(); throw; }
    nd_thread.join();
    if (nd_err) std::rethrow_exception(nd_err);
    S.ordering = "amd";
    S.fill_perm = perm_amd;
    // ~12 us per level (one launch per level) against ~1e11 flop/s and ~1e12 B/s on small fronts
    // round 4: fronts of >= 192 rows and >= 32 pivots run on the matrix cores (a few TFLOP/s), and a level that holds one costs a round of ~6 launches
    // (measured on the wide C3 variant: 230 such levels with 1.1e10 padded flops 45 ms, 17 levels with 3.9e10 flops 11 ms)
    int big_levels_max = 0;
    auto cost_r4 = [&big_levels_max](const Symbolic& X) {
        double t = 12e-6, small_fl = 0.0, big_fl = 0.0;
        int nbig = 0;
        for (int l = 0; l < X.top_nlevels; ++l) {
            bool big = false;
            for (int q = X.top_level_ptr[l]; q < X.top_level_ptr[l + 1] && !big; ++q) {
                const int s2 = X.top_level_sn[q];
                big = big_front(X.front_rows_ptr[s2 + 1] - X.front_rows_ptr[s2], X.sn_first[s2 + 1] - X.sn_first[s2]);
            }
            t += big ? 150e-6 : 12e-6;
            nbig += big;
        }
        for (int s2 = 0; s2 < X.nsuper; ++s2) {
            const double f = X.front_rows_ptr[s2 + 1] - X.front_rows_ptr[s2], w = X.sn_first[s2 + 1] - X.sn_first[s2];
            const double fl = w * f * f - w * w * f + w * w * w / 3.0;
            (big_front((int)f, (int)w) ? big_fl : small_fl) += fl;
        }
        big_levels_max = std::max(big_levels_max, nbig);
        if (debug_token("tree_profile")) std::fprintf(stderr, "cost parts: levels %.2f ms, small-front flops %.3g, big-front flops %.3g, front doubles %.3g\n", 1e3 * t, small_fl, big_fl, (double)X.front_doubles);
        return t + small_fl / 2.5e11 + big_fl / 5e12 + 8.0 * (double)X.front_doubles / 1e12;
    };
    auto cost_r3 = [](const Symbolic& X) { return 12e-6 * (X.top_nlevels + 1) + X.flops / 1e11 + 8.0 * (double)X.front_doubles / 1e12; };
    // the round-3 model stands where it was calibrated (trees with a few levels of big fronts: every frozen fixture keeps its ordering)
    const double c4s = cost_r4(S), c4t = cost_r4(T);
    auto cost = [&](const Symbolic& X) { return big_levels_max > 64 ? (&X == &S ? c4s : c4t) : cost_r3(X); };
    if (debug_token("tree_profile"))
        std::fprintf(stderr, "ordering: amd levels %d (top %d) flops %.3g front doubles %.3g cost %.2f ms (round-3 model %.2f) | nd levels %d (top %d) flops %.3g front doubles %.3g cost %.2f ms (%.2f) | choice %s (round-3 model: %s)\n", S.nlevels, S.top_nlevels,
                     S.flops, (double)S.front_doubles, 1e3 * cost(S), 1e3 * cost_r3(S), T.nlevels, T.top_nlevels, T.flops, (double)T.front_doubles, 1e3 * cost(T), 1e3 * cost_r3(T),
                     cost(T) < 0.7 * cost(S) ? "nd" : "amd", cost_r3(T) < 0.7 * cost_r3(S) ? "nd" : "amd");
    (void)cost_r3;
    if (cost(T) < 0.7 * cost(S)) S = std::move(T);
}

// everything that follows from a fill-reducing ordering perm0 (perm0[new] = old) of K
// forced_first != nullptr: perm0 is already a postorder of its elimination tree and the supernode partition is given (second pass of
// the leaf amalgamation below)
// Bounded fan-in of the assembly tree.  A supernode with very many children (the arrow of a QP with a few dense rows: MM BOYD1 has
// 93 247 single-column leaves under one 18-column root) makes its parent's extend-add -- and the gather of the forward substitution --
// a serial loop over all of them: 67 ms per factorisation where the arithmetic is 5 MFLOP.  Consecutive children are therefore grouped,
// `fanin` at a time, under ACCUMULATOR supernodes without pivot columns (w = 0): their front is the union of the members' update rows (a subset
// of the parent's front), their "factorisation" is the extend-add alone and their update matrix is the whole front.  The kernels need
// nothing new -- every loop over pivots is empty -- and a group of small leaves becomes one workgroup's subtree walk.  Accumulators sit right
// behind their last member, so the numbering stays a postorder with contiguous subtrees; repeated until no fan-in exceeds the bound.
// The partial sums change the order in which a parent receives its children's contributions (fixed, so still reproducible).
static void insert_accumulators(Symbolic& S, int N, int fanin)
{
    for (int pass = 0; pass < 8; ++pass) {
        const int ns = S.nsuper;
        IVec cptr(ns + 1, 0);
        for (int s = 0; s < ns; ++s) if (S.sn_parent[s] >= 0) cptr[S.sn_parent[s] + 1]++;
        for (int s = 0; s < ns; ++s) cptr[s + 1] += cptr[s];
        IVec ch(cptr[ns], 0);
        {
            IVec nx(cptr.begin(), cptr.end() - 1);
            for (int s = 0; s < ns; ++s) if (S.sn_parent[s] >= 0) ch[nx[S.sn_parent[s]]++] = s;
        }
        struct Acc { int parent; int lo, hi; };  // members ch[lo .. hi)
        std::vector<Acc> acc;
        IVec after(ns, -1), under(ns, -1);
        for (int p2 = 0; p2 < ns; ++p2) {
            const int nc = cptr[p2 + 1] - cptr[p2];
            if (nc <= fanin) continue;
            for (int g0 = 0; g0 < nc; g0 += fanin) {
                const int cnt = std::min(fanin, nc - g0);
                if (cnt < 2) continue;
                const int lo = cptr[p2] + g0;
                after[ch[lo + cnt - 1]] = (int)acc.size();
                for (int q = lo; q < lo + cnt; ++q) under[ch[q]] = (int)acc.size();
                acc.push_back({p2, lo, lo + cnt});
            }
        }
        if (acc.empty()) return;
        const int nn = ns + (int)acc.size();
        IVec newid(ns), accid(acc.size());
        {
            int t = 0;
            for (int s = 0; s < ns; ++s) { newid[s] = t++; if (after[s] >= 0) accid[after[s]] = t++; }
        }
        IVec first(nn + 1, N), parent(nn, -1), nind(nn, 1), rptr(nn + 1, 0);
        std::vector<IVec> rows(nn);
        for (int s = 0; s < ns; ++s) {
            const int t = newid[s];
            first[t] = S.sn_first[s];
            rows[t].assign(S.front_rows.begin() + S.front_rows_ptr[s], S.front_rows.begin() + S.front_rows_ptr[s + 1]);
            nind[t] = S.sn_nind[s];
            parent[t] = under[s] >= 0 ? accid[under[s]] : (S.sn_parent[s] >= 0 ? newid[S.sn_parent[s]] : -1);
        }
        for (size_t a = 0; a < acc.size(); ++a) {
            const int t = accid[a];
            const int last_member = ch[acc[a].hi - 1];
            first[t] = S.sn_first[last_member + 1];  // no columns of its own: [first, first)
            parent[t] = newid[acc[a].parent];
            IVec& r = rows[t];
            for (int q = acc[a].lo; q < acc[a].hi; ++q) {
                const int c = ch[q];
                const int w = S.sn_first[c + 1] - S.sn_first[c];
                r.insert(r.end(), S.front_rows.begin() + S.front_rows_ptr[c] + w, S.front_rows.begin() + S.front_rows_ptr[c + 1]);
            }
            std::sort(r.begin(), r.end());
            r.erase(std::unique(r.begin(), r.end()), r.end());
        }
        first[nn] = N;
        S.front_off.assign(nn + 1, 0);
        for (int t = 0; t < nn; ++t) {
            rptr[t + 1] = rptr[t] + (int)rows[t].size();
            S.front_off[t + 1] = S.front_off[t] + (long long)rows[t].size() * (long long)rows[t].size();
        }
        S.front_rows.assign(rptr[nn], 0);
        for (int t = 0; t < nn; ++t) std::copy(rows[t].begin(), rows[t].end(), S.front_rows.begin() + rptr[t]);
        for (int j = 0; j < N; ++j) S.sn_of_col[j] = newid[S.sn_of_col[j]];
        S.sn_first = first; S.sn_parent = parent; S.sn_nind = nind; S.front_rows_ptr = rptr;
        S.nsuper = nn;
        S.front_doubles = S.front_off[nn];
    }
}

static void analyse_with_order(Symbolic& S, const IVec& perm0, const IVec* forced_first)
{
    const int N = S.N;
    IVec pinv0(N);
    for (int i = 0; i < N; ++i) pinv0[perm0[i]] = i;
    if (forced_first) {
        S.P = perm0; S.P_inv = pinv0;
    } else {
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
    // Relaxed amalgamation on top of that: column j may also join when parent(j-1) = j and the explicit zeros this pads into the
    // columns already in the supernode stay small (the front of [a..j] is {a..j} + struct(L_j) either way, because every column of the
    // chain has its parent as the next column).  A supernode costs a fixed few microseconds on the device whatever its size, so tiny
    // ones are merged generously (thresholds in the spirit of CHOLMOD's nrelax / zrelax); fronts are kept <= 64 rows unless exact.
    S.sn_of_col.assign(N, 0);
    S.sn_first.clear();
    if (forced_first) {
        S.sn_first = *forced_first;
        for (size_t q = 0; q < S.sn_first.size(); ++q) {
            const int hi = q + 1 < S.sn_first.size() ? S.sn_first[q + 1] : N;
            for (int j = S.sn_first[q]; j < hi; ++j) S.sn_of_col[j] = (int)q;
        }
    } else {
        const bool relax = true;
        // merged fronts never exceed the largest front of the exact partition (capped at 64): the subtree walkers size their LDS for
        // the largest front, so a bigger one costs occupancy everywhere (measured on the n = 500k chain: 37 -> 50 rows, factor +16 %)
        int fcap = 32;
        {
            int a = 0;
            for (int j = 0; j < N; ++j) {
                const bool joins = j > 0 && S.etree[j - 1] == j && cc[j - 1] == cc[j] + 1;
                if (!joins) a = j;
                if (j + 1 == N || !(S.etree[j] == j + 1 && cc[j] == cc[j + 1] + 1)) fcap = std::max(fcap, (j - a + 1) + cc[j]);
            }
            fcap = std::min(fcap, 64);
        }
        int a = 0;            // first column of the current supernode
        long long sumcc = 0;  // sum of cc over its columns
        for (int j = 0; j < N; ++j) {
            bool joins = false;
            if (j > 0 && S.etree[j - 1] == j) {
                const long long wd = j - a + 1;
                const long long total = wd * (wd - 1) / 2 + wd * cc[j], truth = sumcc + cc[j], Z = total - truth;
                if (Z == 0) joins = true;
                else if (relax && wd + cc[j] <= fcap)
                    joins = wd <= 4 || (wd <= 16 && Z * 10 <= total * 8) || (wd <= 48 && Z * 10 <= total) || Z * 20 <= total;
            }
            if (!joins) { S.sn_first.push_back(j); a = j; sumcc = 0; }
            sumcc += cc[j];
            S.sn_of_col[j] = (int)S.sn_first.size() - 1;
        }
    }
    S.nsuper = (int)S.sn_first.size();
    S.sn_first.push_back(N);
    int ns = S.nsuper;  // (grows when accumulator supernodes are inserted below)
    S.sn_parent.assign(ns, -1);
    S.sn_nind.assign(ns, 1);
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
        {   // columns first .. first + m - 1 are mutually independent iff the parent (= first row of L) of each of them is >= first + m
            const int a0 = S.sn_first[s];
            int m = 1, runmin = S.etree[a0] < 0 ? N : S.etree[a0];
            while (a0 + m <= last) {
                const int e = S.etree[a0 + m] < 0 ? N : S.etree[a0 + m];
                const int rm = std::min(runmin, e);
                if (rm < a0 + m + 1) break;
                runmin = rm; ++m;
            }
            S.sn_nind[s] = m;
        }
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
    // ---- leaf amalgamation.  The KKT matrix of a QP with (block-)diagonal P is full of single-column leaves: a primal variable whose
    // only neighbours are its constraint rows (n = 500k chain: 361 281 of 396 598 supernodes are such leaves, 10-19 of them under every
    // constraint block).  A supernode costs a few microseconds on the device whatever its size, so small childless supernodes are merged
    // into their parent: the merged pivot block is [leaf columns, parent columns] (siblings are structurally independent, their mutual
    // block is padded with zeros), and since struct(L_leaf) is contained in {parent columns} + struct(L_parent) the front is still
    // {pivots} + struct(L_last).  That needs the leaves next to their parent in the elimination order: the children of every supernode
    // are reordered (any postorder of the tree has the same fill) -- other children first, largest subtree last among them so that the
    // chain walks follow the spine, then the merged leaves, then the parent -- and the analysis is redone for that order.
    if (!forced_first) {
        int WMAX = 32, FMAX = 64;
        // Spines of WIDE fronts (band-like problems under AMD: ~1700 supernodes of ~20 pivots and ~370 rows one above the other on the wide C3 variant, every
        // one a tree level = a round of launches of ~45 us; under nested dissection thousands of 2-5 pivot supernodes with 700-row fronts): a supernode
        // takes in the child with the largest subtree while the merged pivot block stays within big_w columns and the padding (the child's columns grow to
        // the merged front) within big_z percent of the merged panel.  The padded panel runs on the matrix cores; a level costs the same launches whatever
        // its width.  Only fronts of 192 rows or more: the trees of the small fixtures, and with them their arithmetic, are untouched.
        int big_w = 256, big_z = 40;
        {   // only DEEP trees: where the levels are few the padding costs more in the substitution than the saved launches give back (CONT-201, 22 levels of
            // big fronts: factorisation 2.02 -> 1.91 ms but 0.94 -> 1.02 ms per solve, two solves per factorisation)
            IVec depth(ns, 0);
            int dmax = 0;
            for (int s2 = 0; s2 < ns; ++s2) {  // children precede parents
                dmax = std::max(dmax, depth[s2]);
                const int ps = S.sn_parent[s2];
                if (ps >= 0) depth[ps] = std::max(depth[ps], depth[s2] + 1);
            }
            if (dmax < 96) big_w = 0;
        }
        if (const char* e = debug_token("big_relax_w")) big_w = std::atoi(e);  // experiments: PIQP_AMD_DEBUG=big_relax_w=<pivots>,big_relax_z=<percent>
        if (const char* e = debug_token("big_relax_z")) big_z = std::atoi(e);
        std::vector<long long> zpad(ns, 0);
        IVec feff(ns, 0);
        IVec weff(ns), live_children(ns, 0), merged_into(ns, -1), sub_cols(ns, 0);
        for (int s2 = 0; s2 < ns; ++s2) { weff[s2] = S.sn_first[s2 + 1] - S.sn_first[s2]; live_children[s2] = S.child_ptr[s2 + 1] - S.child_ptr[s2]; }
        long long nmerged = 0;
        std::vector<std::pair<int, int>> cand;
        for (int p2 = 0; p2 < ns; ++p2) {  // postorder: the children of p2 are final when p2 is visited
            sub_cols[p2] += S.sn_first[p2 + 1] - S.sn_first[p2];
            if (S.sn_parent[p2] >= 0) sub_cols[S.sn_parent[p2]] += sub_cols[p2];
            cand.clear();
            for (int q = S.child_ptr[p2]; q < S.child_ptr[p2 + 1]; ++q) {
                const int c = S.child[q];
                if (live_children[c] == 0 && weff[c] <= 4) cand.push_back({weff[c], c});
            }
            const int last = S.sn_first[p2 + 1] - 1;
            const int fp = (S.sn_first[p2 + 1] - S.sn_first[p2]) + cc[last];  // front of p2 before any merge
            std::sort(cand.begin(), cand.end());
            for (const auto& cw : cand) {
                const int wd = weff[p2] + cw.first;
                const int fr = wd + cc[last];
                if (wd > WMAX) break;
                if (fr > FMAX && fr > fp + 4) break;
                merged_into[cw.second] = p2;
                weff[p2] = wd;
                live_children[p2]--;
                ++nmerged;
            }
            if (big_w > 0) {
                int best = -1;
                for (int q = S.child_ptr[p2]; q < S.child_ptr[p2 + 1]; ++q) {
                    const int c = S.child[q];
                    if (merged_into[c] < 0 && (best < 0 || sub_cols[c] >= sub_cols[best])) best = c;
                }
                if (best >= 0) {
                    const int wd = weff[p2] + weff[best];
                    const int fr = wd + cc[last];
                    if (fr >= 192 && wd <= big_w) {
                        const long long z = zpad[best] + zpad[p2] + (long long)weff[best] * (fr - feff[best]);
                        if (z * 100 <= (long long)wd * fr * big_z) {
                            merged_into[best] = p2;
                            weff[p2] = wd;
                            zpad[p2] = z;
                            live_children[p2] += live_children[best] - 1;
                            ++nmerged;
                        }
                    }
                }
            }
            feff[p2] = weff[p2] + cc[last];
        }
        if (nmerged > 0) {
            // new elimination order: iterative DFS over the supernode tree
            std::vector<IVec> kids(ns), leaves(ns);
            IVec group(ns);  // the supernode whose pivot block s2's columns end up in (parents have larger numbers: resolved top down)
            for (int s2 = ns - 1; s2 >= 0; --s2) group[s2] = merged_into[s2] < 0 ? s2 : group[merged_into[s2]];
            for (int s2 = 0; s2 < ns; ++s2) {
                const int ps = S.sn_parent[s2];
                if (ps < 0) continue;
                if (merged_into[s2] == ps) leaves[ps].push_back(s2); else kids[group[ps]].push_back(s2);  // the live children of a merged member hang under its group
            }
            for (int s2 = 0; s2 < ns; ++s2) {
                if (kids[s2].size() > 1) {  // largest subtree last (ties: keep the original order)
                    size_t best = 0;
                    for (size_t q = 1; q < kids[s2].size(); ++q) if (sub_cols[kids[s2][q]] >= sub_cols[kids[s2][best]]) best = q;
                    const int b = kids[s2][best];
                    kids[s2].erase(kids[s2].begin() + (long)best);
                    kids[s2].push_back(b);
                }
            }
            IVec perm1; perm1.reserve(N);
            IVec first1;
            // emits the pivot block of group g: merged leaves (each possibly a merged group itself), then g's own columns
            std::vector<std::pair<int, int>> st;  // (supernode, phase): phase 0 = descend into live children, 1 = emit the group
            auto emit_cols = [&](int g, auto&& self) -> void {
                for (int l : leaves[g]) self(l, self);
                for (int j = S.sn_first[g]; j < S.sn_first[g + 1]; ++j) perm1.push_back(S.P[j]);
            };
            for (int r = 0; r < ns; ++r) {
                if (S.sn_parent[r] >= 0) continue;
                st.push_back({r, 0});
                while (!st.empty()) {
                    auto [g, ph] = st.back();
                    st.pop_back();
                    if (ph == 0) {
                        st.push_back({g, 1});
                        for (size_t q = kids[g].size(); q-- > 0;) st.push_back({kids[g][q], 0});  // popped in order: first child first
                    } else {
                        first1.push_back((int)perm1.size());
                        emit_cols(g, emit_cols);  // a merged leaf has no live children: the recursion depth is the depth of nested leaf merges (<= WMAX)
                    }
                }
            }
            if ((int)perm1.size() == N) { analyse_with_order(S, perm1, &first1); return; }
        }
    }
    if (!debug_token("no_accumulators")) {
        const int before = S.nsuper;
        insert_accumulators(S, N, 64);
        if (S.nsuper != before) {  // children lists of the extended tree
            const int nn = S.nsuper;
            S.child_ptr.assign(nn + 1, 0);
            for (int s2 = 0; s2 < nn; ++s2) if (S.sn_parent[s2] >= 0) S.child_ptr[S.sn_parent[s2] + 1]++;
            for (int s2 = 0; s2 < nn; ++s2) S.child_ptr[s2 + 1] += S.child_ptr[s2];
            S.child.assign(S.child_ptr[nn], 0);
            IVec nx(S.child_ptr.begin(), S.child_ptr.end() - 1);
            for (int s2 = 0; s2 < nn; ++s2) if (S.sn_parent[s2] >= 0) S.child[nx[S.sn_parent[s2]]++] = s2;
            ns = nn;
        }
    }
    if (debug_token("sn_stats")) {
        long long leaves = 0, leaf_small = 0, one_child = 0, hist[9] = {0};
        long long nchild_hist[6] = {0};
        for (int s = 0; s < ns; ++s) {
            const int nc = S.child_ptr[s + 1] - S.child_ptr[s], w = S.sn_first[s + 1] - S.sn_first[s];
            hist[std::min(w, 8)]++;
            nchild_hist[std::min(nc, 5)]++;
            if (nc == 0) { leaves++; if (w <= 4) leaf_small++; }
            if (nc == 1) one_child++;
        }
        std::fprintf(stderr, "supernodes %d: leaves %lld (w<=4: %lld), one child %lld; children histogram 0..5+: %lld %lld %lld %lld %lld %lld; width histogram 1..8+: %lld %lld %lld %lld %lld %lld %lld %lld\n",
                     ns, leaves, leaf_small, one_child, nchild_hist[0], nchild_hist[1], nchild_hist[2], nchild_hist[3], nchild_hist[4], nchild_hist[5], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7], hist[8]);
    }
    IVec level(ns, 0);
    int maxlev = 0;
    for (int s = 0; s < ns; ++s) {  // children precede parents (postorder)
        const int ps = S.sn_parent[s];
        if (ps >= 0) level[ps] = std::max(level[ps], level[s] + 1);
        maxlev = std::max(maxlev, level[s]);
    }
    S.nlevels = ns ? maxlev + 1 : 0;
    if (debug_token("tree_profile")) {  // debugging aid: fronts per level band, their pivots and orders
        std::vector<long long> cnt(S.nlevels, 0), wsum(S.nlevels, 0), fmaxl(S.nlevels, 0), fsum(S.nlevels, 0);
        for (int s = 0; s < ns; ++s) {
            const int l = level[s], w = S.sn_first[s + 1] - S.sn_first[s], f = S.front_rows_ptr[s + 1] - S.front_rows_ptr[s];
            cnt[l]++; wsum[l] += w; fsum[l] += f; fmaxl[l] = std::max<long long>(fmaxl[l], f);
        }
        const int step = std::max(1, S.nlevels / 40);
        for (int l = 0; l < S.nlevels; l += step) {
            long long c = 0, w = 0, f = 0, fm = 0;
            for (int q = l; q < std::min(S.nlevels, l + step); ++q) { c += cnt[q]; w += wsum[q]; f += fsum[q]; fm = std::max(fm, fmaxl[q]); }
            std::fprintf(stderr, "levels %5d..%5d: fronts %7lld  mean pivots %6.2f  mean order %7.1f  max order %5lld\n", l, std::min(S.nlevels, l + step) - 1, c, (double)w / c, (double)f / c, fm);
        }
    }
    S.level_ptr.assign(S.nlevels + 1, 0);
    for (int s = 0; s < ns; ++s) S.level_ptr[level[s] + 1]++;
    for (int l = 0; l < S.nlevels; ++l) S.level_ptr[l + 1] += S.level_ptr[l];
    S.level_sn.assign(ns, 0);
    {
        IVec nx(S.level_ptr.begin(), S.level_ptr.end() - 1);
        for (int s = 0; s < ns; ++s) S.level_sn[nx[level[s]]++] = s;
    }

    // ---- subtree-to-workgroup partition: a supernode roots a "small subtree" when its whole subtree has at most SUB_COLS
    // columns and its parent's does not.  Supernodes are numbered in postorder, so a subtree is the contiguous range
    // [s - desc[s], s].
    auto subtree_schedule = [&](int SUB_COLS, int SUB_FMAX, IVec& sub_lo, IVec& sub_hi, int& sub_max_front, IVec& top_level_ptr, IVec& top_level_sn, int& top_nlevels) {
        IVec cols(ns, 0), desc(ns, 0), fmax(ns, 0);
        for (int s = 0; s < ns; ++s) {
            cols[s] += S.sn_first[s + 1] - S.sn_first[s];
            fmax[s] = std::max(fmax[s], S.front_rows_ptr[s + 1] - S.front_rows_ptr[s]);
            const int ps = S.sn_parent[s];
            if (ps >= 0) { cols[ps] += cols[s]; desc[ps] += desc[s] + 1; fmax[ps] = std::max(fmax[ps], fmax[s]); }
        }
        for (int s = 0; s < ns; ++s) if (fmax[s] > SUB_FMAX) cols[s] = SUB_COLS + 1;  // a subtree holding a wide front is not "small"
        IVec in_sub(ns, 0);
        sub_lo.clear(); sub_hi.clear(); sub_max_front = 0;
        for (int s = 0; s < ns; ++s) {
            const int ps = S.sn_parent[s];
            if (cols[s] <= SUB_COLS && (ps < 0 || cols[ps] > SUB_COLS)) {
                sub_lo.push_back(s - desc[s]); sub_hi.push_back(s);
                for (int t = s - desc[s]; t <= s; ++t) { in_sub[t] = 1; sub_max_front = std::max(sub_max_front, S.front_rows_ptr[t + 1] - S.front_rows_ptr[t]); }
            }
        }
        IVec tl(ns, -1);
        int tmax = -1;
        for (int s = 0; s < ns; ++s) {
            if (in_sub[s]) continue;
            if (tl[s] < 0) tl[s] = 0;
            tmax = std::max(tmax, tl[s]);
            const int ps = S.sn_parent[s];
            if (ps >= 0) tl[ps] = std::max(tl[ps], tl[s] + 1);
        }
        // a parent of a subtree root starts at level 0 (its subtree children are complete before the level phase)
        top_nlevels = tmax + 1;
        top_level_ptr.assign(top_nlevels + 1, 0);
        for (int s = 0; s < ns; ++s) if (!in_sub[s]) top_level_ptr[tl[s] + 1]++;
        for (int l = 0; l < top_nlevels; ++l) top_level_ptr[l + 1] += top_level_ptr[l];
        top_level_sn.assign(top_level_ptr[top_nlevels], 0);
        IVec nx(top_level_ptr.begin(), top_level_ptr.end() - 1);
        for (int s = 0; s < ns; ++s) if (!in_sub[s]) top_level_sn[nx[tl[s]]++] = s;
    };
    {
        int SUB_COLS = 192, SUB_FMAX = 96;  // two 96 x 96 fronts fit the 160 KiB of LDS of one workgroup
        if (const char* e = debug_token("sub_cols")) SUB_COLS = std::max(8, std::atoi(e));  // bitwise-consistency test: smaller walks
        subtree_schedule(SUB_COLS, SUB_FMAX, S.sub_lo, S.sub_hi, S.sub_max_front, S.top_level_ptr, S.top_level_sn, S.top_nlevels);
        // far more subtrees than the device runs at a time (a few thousand walks): longer walks instead, which also keeps the wide low
        // levels of the tree out of the level launches (n = 500k chain: 5875 -> 2900 subtrees, factor 1.03 -> 0.91 ms)
        if (!debug_token("sub_cols"))
            while ((int)S.sub_lo.size() > 4096 && SUB_COLS < 768) {
                SUB_COLS *= 2;
                subtree_schedule(SUB_COLS, SUB_FMAX, S.sub_lo, S.sub_hi, S.sub_max_front, S.top_level_ptr, S.top_level_sn, S.top_nlevels);
            }
        S.nsub = (int)S.sub_lo.size();
        // substitution: one wave walks a subtree front by front, so the sweep lasts as long as the longest walk -- shorter walks and a
        // larger flag-ordered top are faster there (C3 backend solve 0.52 -> 0.34 ms, C5-size chain 0.89 -> 0.79 ms at 24 columns, chains of the top merged into walks), the factorisation prefers the
        // long LDS-resident walks above
        int SOLVE_COLS = 24;
        if (const char* e = debug_token("solve_sub_cols")) SOLVE_COLS = std::atoi(e);
        if (SOLVE_COLS <= 0 || SOLVE_COLS >= SUB_COLS) {
            S.solve_sub_lo = S.sub_lo; S.solve_sub_hi = S.sub_hi; S.solve_top_level_ptr = S.top_level_ptr; S.solve_top_level_sn = S.top_level_sn;
            S.solve_top_nlevels = S.top_nlevels; S.solve_sub_max_front = S.sub_max_front;
        } else {
            subtree_schedule(std::max(8, SOLVE_COLS), SUB_FMAX, S.solve_sub_lo, S.solve_sub_hi, S.solve_sub_max_front, S.solve_top_level_ptr, S.solve_top_level_sn, S.solve_top_nlevels);
        }
        {   // chains of the substitution's top
            const int nt = (int)S.solve_top_level_sn.size();
            IVec pos(ns, -1);
            for (int q = 0; q < nt; ++q) pos[S.solve_top_level_sn[q]] = q;
            const bool merge = !debug_token("solve_no_chains");
            std::vector<std::pair<int, std::pair<int, int>>> walks;  // (position of the last supernode, (lo, hi))
            int s = 0;
            while (s < ns) {
                if (pos[s] < 0) { ++s; continue; }
                int hi = s;
                while (merge && hi + 1 < ns && pos[hi + 1] >= 0 && S.sn_parent[hi] == hi + 1) ++hi;
                walks.push_back({pos[hi], {s, hi}});
                s = hi + 1;
            }
            std::sort(walks.begin(), walks.end());
            S.solve_walk_lo.clear(); S.solve_walk_hi.clear();
            for (const auto& wk : walks) { S.solve_walk_lo.push_back(wk.second.first); S.solve_walk_hi.push_back(wk.second.second); }
        }
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

    // ---- per-supernode lists of the K entries of its front (value index, offset inside the front): lets a workgroup assemble
    // a front straight into LDS
    {
        S.fe_ptr.assign(ns + 1, 0);
        const int nzk = S.Cp[N];
        IVec owner(nzk);
        for (int k = 0; k < N; ++k) for (int q = S.Cp[k]; q < S.Cp[k + 1]; ++q) { owner[q] = S.sn_of_col[S.Ci[q]]; S.fe_ptr[owner[q] + 1]++; }
        for (int s = 0; s < ns; ++s) S.fe_ptr[s + 1] += S.fe_ptr[s];
        S.fe_q.assign(nzk, 0); S.fe_off.assign(nzk, 0);
        IVec nx(S.fe_ptr.begin(), S.fe_ptr.end() - 1);
        for (int q = 0; q < nzk; ++q) { const int t = nx[owner[q]]++; S.fe_q[t] = q; S.fe_off[t] = (int)(S.a_dst[q] - S.front_off[owner[q]]); }
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

// ---- partition_tree --------------------------------------------------------------------------------------------------------
// Subtree-to-rank mapping: start from the roots of the assembly forest and keep replacing the heaviest candidate subtree by
// its children (its root joins the shared top) until no expandable candidate carries more than 1 / (8 world) of the work.
// Workgroup subtrees (sub_lo..sub_hi) are atomic.  Candidates are then dealt to the ranks in postorder, contiguous and
// balanced by prefix sums of their work, so that every rank owns one contiguous range of the (postordered) elimination order
// interrupted only by shared supernodes: for a multistage chain that is a contiguous range of stages.
void partition_tree(const Symbolic& S, int world, Partition& P)
{
    const int ns = S.nsuper;
    P = Partition();
    P.world = world;
    P.owner.assign(ns, -1);
    P.work.assign(world, 0.0);
    P.span_lo.assign(world, 0); P.span_hi.assign(world, 0);
    if (ns == 0) { P.bmat_off.assign(1, 0); P.bvec_off.assign(1, 0); return; }
    IVec desc(ns, 0), in_sub(ns, 0);
    DVec sw(ns, 0.0);
    for (int k = 0; k < S.nsub; ++k) for (int t = S.sub_lo[k]; t <= S.sub_hi[k]; ++t) in_sub[t] = 1;
    for (int s = 0; s < ns; ++s) {
        const double w = S.sn_first[s + 1] - S.sn_first[s], f = S.front_rows_ptr[s + 1] - S.front_rows_ptr[s], u = f - w;
        sw[s] += w * w * w / 3.0 + w * w * u + w * u * u + 4.0 * f * w + 2000.0;  // factor + substitution + a per-front latency term
        const int ps = S.sn_parent[s];
        if (ps >= 0) {
            if (ps <= s) throw std::runtime_error("partition_tree: supernodes are not in postorder");
            sw[ps] += sw[s]; desc[ps] += desc[s] + 1;
        }
    }
    for (int s = 0; s < ns; ++s) if (S.sn_parent[s] < 0) P.total_work += sw[s];
    // candidates kept in a max-heap on subtree work; only top supernodes (not inside a workgroup subtree) can be expanded
    auto cmp = [&](int a, int b) { return sw[a] < sw[b] || (sw[a] == sw[b] && a < b); };
    std::vector<int> heap, fixed;
    for (int s = 0; s < ns; ++s) if (S.sn_parent[s] < 0) heap.push_back(s);
    std::make_heap(heap.begin(), heap.end(), cmp);
    const double limit = P.total_work / (8.0 * world);
    std::vector<char> shared(ns, 0);
    while (!heap.empty() && world > 1) {
        std::pop_heap(heap.begin(), heap.end(), cmp);
        const int s = heap.back(); heap.pop_back();
        const bool expandable = !in_sub[s] && S.child_ptr[s + 1] > S.child_ptr[s];
        if (sw[s] <= limit || (int)(heap.size() + fixed.size()) >= 64 * world) { fixed.push_back(s); fixed.insert(fixed.end(), heap.begin(), heap.end()); heap.clear(); break; }
        if (!expandable) { fixed.push_back(s); continue; }
        shared[s] = 1;
        for (int ci = S.child_ptr[s]; ci < S.child_ptr[s + 1]; ++ci) { heap.push_back(S.child[ci]); std::push_heap(heap.begin(), heap.end(), cmp); }
    }
    fixed.insert(fixed.end(), heap.begin(), heap.end());
    std::sort(fixed.begin(), fixed.end());
    double cand_work = 0.0;
    for (int c : fixed) cand_work += sw[c];
    for (int s = 0; s < ns; ++s) if (shared[s]) {
        double own = sw[s];
        for (int ci = S.child_ptr[s]; ci < S.child_ptr[s + 1]; ++ci) own -= sw[S.child[ci]];
        P.shared_work += own;
    }
    // contiguous deal: candidate c goes to the rank in whose work interval the midpoint of c falls
    {
        double acc = 0.0;
        std::vector<char> seen(world, 0);
        for (int c : fixed) {
            int r = world == 1 ? 0 : (int)((acc + 0.5 * sw[c]) / cand_work * world);
            r = std::min(std::max(r, 0), world - 1);
            acc += sw[c];
            for (int t = c - desc[c]; t <= c; ++t) P.owner[t] = r;
            P.work[r] += sw[c];
            const int lo = S.sn_first[c - desc[c]], hi = S.sn_first[c + 1];
            if (!seen[r]) { P.span_lo[r] = lo; seen[r] = 1; }
            P.span_hi[r] = hi;
        }
        for (int r = 0; r < world; ++r) P.max_span = std::max(P.max_span, P.span_hi[r] - P.span_lo[r]);
    }
    P.bmat_off.assign(1, 0); P.bvec_off.assign(1, 0);
    for (int s = 0; s < ns; ++s) {
        const int ps = S.sn_parent[s];
        if (P.owner[s] >= 0 && ps >= 0 && P.owner[ps] < 0) {
            const long long u = (S.front_rows_ptr[s + 1] - S.front_rows_ptr[s]) - (S.sn_first[s + 1] - S.sn_first[s]);
            P.boundary.push_back(s);
            P.bmat_off.push_back(P.bmat_off.back() + u * u);
            P.bvec_off.push_back(P.bvec_off.back() + (int)u);
        }
        if (P.owner[s] >= 0 && ps >= 0 && P.owner[ps] >= 0 && P.owner[ps] != P.owner[s]) throw std::runtime_error("partition_tree: a parent and its child are owned by different ranks");
        if (P.owner[s] < 0 && ps >= 0 && P.owner[ps] >= 0) throw std::runtime_error("partition_tree: a shared supernode below an owned one");
    }
}

// ---- the reference's elimination replayed symbolically (round 5; see UpLooking in the header)
void analyse_uplooking(const Symbolic& S, const pq_sparse_data* d, UpLooking& U)
{
    const int N = S.N;
    U.n = S.n; U.p = S.p; U.m = S.m; U.N = N; U.mode = S.mode;
    U.perm.assign(N, 0); U.perm_inv.assign(N, 0);
    if (N > 0) amd_order(N, S.Kp.data(), S.Ki.data(), U.perm.data());   // sparse/ordering.hpp:67-84
    for (int k = 0; k < N; ++k) U.perm_inv[U.perm[k]] = k;
    permute_sym_upper(N, S.Kp, S.Ki, U.perm_inv.data(), U.Cp, U.Ci, U.PKi);  // sparse/utils.hpp:32-128
    U.diag_pos.assign(N, 0);
    for (int col = 0; col < N; ++col) U.diag_pos[col] = U.Cp[U.perm_inv[col] + 1] - 1;  // kkt_full.hpp:181: PKPt.valuePtr()[PKPt.outerIndexPtr()[ordering.inv(col) + 1] - 1]
    for (int col = 0; col < N; ++col)
        if (U.Ci[U.diag_pos[col]] != U.perm_inv[col]) throw std::runtime_error("up-looking analysis: a column of P K P' does not end in its diagonal");
    const bool eq = (S.mode & 1) != 0, ineq = (S.mode & 2) != 0;
    const int nzP = d->P_colptr[S.n], nzA = (S.p && !eq) ? d->AT_colptr[S.p] : 0, nzG = (S.m && !ineq) ? d->GT_colptr[S.m] : 0;
    U.mapP.resize(nzP); U.mapA.resize(nzA); U.mapG.resize(nzG);
    for (int q = 0; q < nzP; ++q) U.mapP[q] = U.PKi[S.P_utri_to_Ki[q]];
    for (int q = 0; q < nzA; ++q) U.mapA[q] = U.PKi[S.AT_to_Ki[q]];
    for (int q = 0; q < nzG; ++q) U.mapG[q] = U.PKi[S.GT_to_Ki[q]];
    // LDLt::factorize_symbolic_upper_triangular (ldlt.hpp:42-99)
    IVec colcount;
    elimination_tree(N, U.Cp, U.Ci, U.etree, colcount);
    U.Lp.assign(N + 1, 0);
    for (int k = 0; k < N; ++k) U.Lp[k + 1] = U.Lp[k] + colcount[k];
    const int nnzL = U.Lp[N];
    U.nnzL = nnzL;
    U.Li.assign(nnzL, 0); U.Lcol.assign(nnzL, 0);
    U.Rp.assign(N + 1, 0); U.Rcol.assign(nnzL, 0); U.Rpos.assign(nnzL, 0);
    // the pattern walk of factorize_numeric_upper_triangular (ldlt.hpp:121-143) with the bookkeeping of :159-161: row k's entries in the order the
    // reference visits them, and where each lands in L's columns (a column receives its entries in ascending row order)
    IVec flag(N, -1), pattern(N), lnz(N, 0);
    int w = 0;
    for (int k = 0; k < N; ++k) {
        int top = N;
        flag[k] = k;
        for (int p = U.Cp[k]; p < U.Cp[k + 1]; ++p) {
            int i = U.Ci[p];
            int len = 0;
            for (; flag[i] != k; i = U.etree[i]) { pattern[len++] = i; flag[i] = k; }
            while (len > 0) pattern[--top] = pattern[--len];
        }
        U.Rp[k] = w;
        for (; top < N; ++top) {
            const int i = pattern[top];
            const int pos = U.Lp[i] + lnz[i]++;
            U.Li[pos] = k; U.Lcol[pos] = i;
            U.Rcol[w] = i; U.Rpos[w] = pos;
            ++w;
        }
    }
    U.Rp[N] = w;
    if (w != nnzL) throw std::runtime_error("up-looking analysis: row and column counts of L disagree");
    U.flops = 0.0;
    for (int k = 0; k < N; ++k) U.flops += (double)colcount[k] * colcount[k] + 3.0 * colcount[k];
    IVec h(N, 0);
    std::vector<long long> cp(N, 0);
    U.height = 0; U.crit_steps = 0;
    for (int k = 0; k < N; ++k) {
        h[k] += 1; cp[k] += U.Rp[k + 1] - U.Rp[k];
        U.height = std::max(U.height, h[k]); U.crit_steps = std::max(U.crit_steps, cp[k]);
        const int pa = U.etree[k];
        if (pa >= 0) { h[pa] = std::max(h[pa], h[k]); cp[pa] = std::max(cp[pa], cp[k]); }
    }
    // ---- schedule: paths of the elimination tree.  Every row continues the path of its HEAVIEST child (the one with the longest chain of dependent
    // steps below it) until the path holds UL_PATH rows; a path is one task.  Row k of a task is computed in two passes (sparse_exact.hip):
    //   row pass   (any wave, all rows of a task concurrently): the entries of row k whose column lies OUTSIDE the task, exactly as the reference's loop
    //              forms them, except that the updates these columns send to rows of the task's own path are left out;
    //   path pass  (one wave per task, its rows one after the other): those left-out updates and the entries in the task's own columns, with the path rows
    //              as lanes -- every entry still receives its terms in the reference's order (an update into path row c comes from a descendant of c, which
    //              the reference's pattern order visits before c; the pass walks the pattern of row k once, in that order).
    constexpr int UL_PATH = 64;
    IVec heavy(N, -1);
    {
        std::vector<long long> best(N, -1);
        for (int k = 0; k < N; ++k) { const int pa = U.etree[k]; if (pa >= 0 && cp[k] > best[pa]) { best[pa] = cp[k]; heavy[pa] = k; } }
    }
    U.row_task.assign(N, -1); U.row_lane.assign(N, 0); U.row_prev.assign(N, -1);
    std::vector<IVec> rows_of;
    for (int k = 0; k < N; ++k) {
        const int hc = heavy[k];
        if (hc >= 0 && (int)rows_of[U.row_task[hc]].size() < UL_PATH && rows_of[U.row_task[hc]].back() == hc) {
            const int t = U.row_task[hc];
            U.row_task[k] = t; U.row_lane[k] = (int)rows_of[t].size(); U.row_prev[k] = hc;
            rows_of[t].push_back(k);
        } else {
            U.row_task[k] = (int)rows_of.size(); U.row_lane[k] = 0;
            rows_of.push_back(IVec(1, k));
        }
    }
    const int nt = (int)rows_of.size();
    U.task_ptr.assign(nt + 1, 0);
    U.task_rows.clear();
    for (int t = 0; t < nt; ++t) { U.task_rows.insert(U.task_rows.end(), rows_of[t].begin(), rows_of[t].end()); U.task_ptr[t + 1] = (int)U.task_rows.size(); }
    // what the row pass of k waits for: the children outside the task of k AND of the rows below it on the task's path (complete rows: every column the pass reads
    // is then final).  The lists of a task's rows are prefixes of one another; they are stored per row so that a pass polls all its words at once (a chain of
    // "the row before me has been released" words cost two microseconds per link).
    U.dep_ptr.assign(N + 1, 0);
    U.dep.clear();
    {
        std::vector<IVec> kids(N);
        for (int k = 0; k < N; ++k) { const int pa = U.etree[k]; if (pa >= 0 && U.row_task[pa] != U.row_task[k]) kids[pa].push_back(k); }
        for (int k = 0; k < N; ++k) {
            const IVec& R = rows_of[U.row_task[k]];
            for (int c = 0; c <= U.row_lane[k]; ++c) U.dep.insert(U.dep.end(), kids[R[c]].begin(), kids[R[c]].end());
            U.dep_ptr[k + 1] = (int)U.dep.size();
        }
    }
    // tickets: rows ascending; the path pass of a task right behind the row pass of its last row (every wait is for an earlier ticket)
    U.tk_kind.clear(); U.tk_id.clear();
    for (int k = 0; k < N; ++k) {
        U.tk_kind.push_back(0); U.tk_id.push_back(k);
        const int t = U.row_task[k];
        if (rows_of[t].size() > 1 && rows_of[t].back() == k) { U.tk_kind.push_back(1); U.tk_id.push_back(t); }
    }
    // per entry: how many leading entries of its column the row pass scatters (-1: a column of the task's own path, left to the path pass), and the
    // row of the task's table the entry's column has; per task: the table itself -- one row per column that reaches the path (outside columns, then the path's
    // own), one slot per path row: the values L(path row, column) the path pass reads (written by the passes that produce them) and their presence bits
    U.Rcnt.assign(nnzL, 0); U.Rtab.assign(nnzL, 0);
    U.tab_ptr.assign(nt + 1, 0); U.mask_ptr.assign(nt + 1, 0); U.task_nU.assign(nt, 0);
    U.Tmask.clear();
    {
        IVec ucol(N, -1);  // column -> table row within the task being built
        for (int t = 0; t < nt; ++t) {
            const IVec& R = rows_of[t];
            const int W = (int)R.size();
            U.tab_ptr[t + 1] = U.tab_ptr[t]; U.mask_ptr[t + 1] = U.mask_ptr[t];
            IVec ucols;
            for (int k : R)
                for (int e = U.Rp[k]; e < U.Rp[k + 1]; ++e) {
                    const int i = U.Rcol[e];
                    if (U.row_task[i] == t) { U.Rcnt[e] = -1; continue; }
                    int q = U.Lp[i];
                    while (U.row_task[U.Li[q]] != t) ++q;   // (ends: row k itself is an entry of column i)
                    U.Rcnt[e] = q - U.Lp[i];
                    if (ucol[i] < 0) { ucol[i] = (int)ucols.size(); ucols.push_back(i); }
                }
            const int nU = (int)ucols.size();
            U.task_nU[t] = nU;
            const size_t m0 = U.Tmask.size();
            U.Tmask.resize(m0 + nU + W, 0ull);
            for (int u = 0; u < nU + W; ++u) {
                const int col = u < nU ? ucols[u] : R[u - nU];
                for (int q = U.Lp[col]; q < U.Lp[col + 1]; ++q) if (U.row_task[U.Li[q]] == t) U.Tmask[m0 + u] |= 1ull << U.row_lane[U.Li[q]];
            }
            for (int k : R)
                for (int e = U.Rp[k]; e < U.Rp[k + 1]; ++e) { const int i = U.Rcol[e]; U.Rtab[e] = U.row_task[i] == t ? nU + U.row_lane[i] : ucol[i]; }
            for (int i : ucols) ucol[i] = -1;
            U.tab_ptr[t + 1] = U.tab_ptr[t] + (nU + W) * W;
            U.mask_ptr[t + 1] = (int)U.Tmask.size();
        }
    }
    // ---- schedule of the substitution (lsolve / ltsolve, ldlt.hpp:171-218) on the same tasks.  Forward: a task's rows as lanes, its table rows (columns that reach
    // it) walked in ascending column order -- x_t loses fl(L(t, j) x_j) for j ascending, as the reference's column loop delivers them; a task waits for the tasks
    // that end in a child of one of its rows.  Backward: a task's columns from the last to the first, each column's entries in ascending row order; a task waits for
    // the task of its last row's parent.  Tasks are taken in the order of their last rows (forward ascending, backward descending).
    {
        U.fs_ptr.assign(nt + 1, 0); U.fs_u.clear(); U.fs_col.clear();
        U.tdep_ptr.assign(nt + 1, 0); U.tdep.clear(); U.tparent.assign(nt, -1);
        std::vector<std::pair<int, int>> srt;
        IVec ucol_of;  // table row -> column, rebuilt per task the way the table was
        IVec seen(N, -1);
        for (int t = 0; t < nt; ++t) {
            const IVec& R = rows_of[t];
            const int W = (int)R.size(), nU = U.task_nU[t];
            ucol_of.assign(nU + W, -1);
            for (int k : R)
                for (int e = U.Rp[k]; e < U.Rp[k + 1]; ++e) ucol_of[U.Rtab[e]] = U.Rcol[e];
            for (int c = 0; c < W; ++c) ucol_of[nU + c] = R[c];
            srt.clear();
            for (int u = 0; u < nU + W; ++u) srt.emplace_back(ucol_of[u], u);
            std::sort(srt.begin(), srt.end());
            for (const auto& pr : srt) { U.fs_col.push_back(pr.first); U.fs_u.push_back(pr.second); }
            U.fs_ptr[t + 1] = (int)U.fs_u.size();
            for (int q = U.dep_ptr[R.back()]; q < U.dep_ptr[R.back() + 1]; ++q) U.tdep.push_back(U.row_task[U.dep[q]]);  // (the last row's list is the union over the task)
            U.tdep_ptr[t + 1] = (int)U.tdep.size();
            const int pa = U.etree[R.back()];
            U.tparent[t] = pa >= 0 ? U.row_task[pa] : -1;
        }
        std::vector<std::pair<int, int>> ord;
        for (int t = 0; t < nt; ++t) ord.emplace_back(rows_of[t].back(), t);
        std::sort(ord.begin(), ord.end());
        U.tsort.clear();
        for (const auto& pr : ord) U.tsort.push_back(pr.second);
        // per CSC entry: the lane of its row when that row lies on the path of its column's task (the backward sweep then takes x from that lane), else -1
        U.Lsrc.assign(nnzL, -1);
        for (int j = 0; j < N; ++j)
            for (int q = U.Lp[j]; q < U.Lp[j + 1]; ++q) if (U.row_task[U.Li[q]] == U.row_task[j]) U.Lsrc[q] = U.row_lane[U.Li[q]];
        (void)seen;
    }
}

}  // namespace sparse
}  // namespace pq
