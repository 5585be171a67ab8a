// Bandwidth-reducing symmetric reordering at ingest (reverse Cuthill-McKee) for matrices whose rows come in an order
// that scatters the x gathers of the SpMV over the whole vector.
//
// The reference's operators accept any sparsity pattern (MatOp/SparseSymMatProd.h:83-88, SparseGenMatProd.h:82-87) and a
// CPU's cache hierarchy hides much of a bad ordering; on the GPU every x[col] gather that misses the 4 MiB per-XCD L2
// pulls a whole 128-byte line through the fabric, so a stencil matrix in random order runs at a tenth of the bandwidth
// of the same matrix in banded order.  Reordering is pure preprocessing (integer work on the pattern): the solver then
// works on P A P' and un-permutes what it hands back; eigenvalues are unchanged.
//
// Host code (one thread, once per matrix):  O(nnz) breadth-first searches + a sort of every adjacency list by degree.
//   1. pattern of A + A' as an adjacency structure (the symmetric operators are already structurally symmetric);
//   2. per connected component: pseudo-peripheral start vertex (George & Liu: repeat BFS from a minimum-degree vertex
//      of the last level until the eccentricity stops growing), Cuthill-McKee numbering (BFS, neighbours in order of
//      increasing degree), reversed at the end;
//   3. early exit: if the widest BFS level of the first search is a sizeable fraction of the component (expander-like
//      graphs, e.g. uniformly random columns), no ordering can localise the gathers and the search stops there.
#include "reorder.hpp"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <numeric>

namespace mispec {

namespace {

struct Graph
{
    int64_t n = 0;
    std::vector<int64_t> ptr;  // n + 1
    std::vector<int32_t> adj;  // neighbours, self loops removed, duplicates removed
    int32_t degree(int32_t v) const { return int32_t(ptr[size_t(v) + 1] - ptr[size_t(v)]); }
};

// adjacency of the pattern of A + A' (n x n, CSR with int32 offsets local to the arrays given)
Graph build_graph(int64_t n, const int32_t* rowptr, const int32_t* colind, bool symmetric_pattern)
{
    Graph g;
    g.n = n;
    g.ptr.assign(size_t(n) + 1, 0);
    if (symmetric_pattern)
    {
        for (int64_t i = 0; i < n; i++)
            for (int32_t p = rowptr[i]; p < rowptr[i + 1]; p++)
                if (colind[p] != i)
                    g.ptr[size_t(i) + 1]++;
        for (int64_t i = 0; i < n; i++)
            g.ptr[size_t(i) + 1] += g.ptr[size_t(i)];
        g.adj.resize(size_t(g.ptr[size_t(n)]));
        for (int64_t i = 0; i < n; i++)
        {
            int64_t q = g.ptr[size_t(i)];
            for (int32_t p = rowptr[i]; p < rowptr[i + 1]; p++)
                if (colind[p] != i)
                    g.adj[size_t(q++)] = colind[p];
        }
        return g;
    }
    // general pattern: union with the transpose, then sort + unique per vertex
    for (int64_t i = 0; i < n; i++)
        for (int32_t p = rowptr[i]; p < rowptr[i + 1]; p++)
            if (colind[p] != i)
            {
                g.ptr[size_t(i) + 1]++;
                g.ptr[size_t(colind[p]) + 1]++;
            }
    for (int64_t i = 0; i < n; i++)
        g.ptr[size_t(i) + 1] += g.ptr[size_t(i)];
    std::vector<int32_t> tmp(size_t(g.ptr[size_t(n)]));
    std::vector<int64_t> fill(g.ptr.begin(), g.ptr.end() - 1);
    for (int64_t i = 0; i < n; i++)
        for (int32_t p = rowptr[i]; p < rowptr[i + 1]; p++)
            if (colind[p] != i)
            {
                tmp[size_t(fill[size_t(i)]++)] = colind[p];
                tmp[size_t(fill[size_t(colind[p])]++)] = int32_t(i);
            }
    std::vector<int64_t> nptr(size_t(n) + 1, 0);
    g.adj.reserve(tmp.size());
    for (int64_t i = 0; i < n; i++)
    {
        auto b = tmp.begin() + g.ptr[size_t(i)], e = tmp.begin() + g.ptr[size_t(i) + 1];
        std::sort(b, e);
        e = std::unique(b, e);
        g.adj.insert(g.adj.end(), b, e);
        nptr[size_t(i) + 1] = int64_t(g.adj.size());
    }
    g.ptr.swap(nptr);
    return g;
}

// BFS from `start` over unvisited (mark != stamp ... we use level[] = -1 as "unvisited in this search") vertices of one
// component.  Returns the number of levels; order[] receives the vertices in BFS order (count written to *count), the
// last level's range is [last_begin, count).
struct Bfs
{
    std::vector<int32_t> order;
    std::vector<int32_t> level;  // -1 = not reached in the current search
    int64_t count = 0, last_begin = 0, widest = 0;
    int nlevels = 0;
};

void bfs(const Graph& g, int32_t start, const std::vector<uint8_t>& done, Bfs& s)
{
    // reset only what the previous search of this component touched
    for (int64_t i = 0; i < s.count; i++)
        s.level[size_t(s.order[size_t(i)])] = -1;
    s.count = 0;
    s.order[size_t(s.count++)] = start;
    s.level[size_t(start)] = 0;
    int64_t head = 0, level_begin = 0;
    s.nlevels = 0;
    s.widest = 1;
    while (head < s.count)
    {
        const int64_t level_end = s.count;
        level_begin = head;
        for (; head < level_end; head++)
        {
            const int32_t v = s.order[size_t(head)];
            for (int64_t p = g.ptr[size_t(v)]; p < g.ptr[size_t(v) + 1]; p++)
            {
                const int32_t u = g.adj[size_t(p)];
                if (s.level[size_t(u)] < 0 && !done[size_t(u)])
                {
                    s.level[size_t(u)] = s.nlevels + 1;
                    s.order[size_t(s.count++)] = u;
                }
            }
        }
        s.widest = std::max(s.widest, level_end - level_begin);
        s.nlevels++;
    }
    s.last_begin = level_begin;
}

}  // namespace

double far_fraction(int64_t n, const int32_t* rowptr, const int32_t* colind, const int32_t* inv, int64_t window, int64_t row0)
{
    rowptr += row0;
    const int nt = int(std::max<int64_t>(1, std::min<int64_t>(ingest_threads(), n / 65536)));
    std::vector<int64_t> far(static_cast<size_t>(nt), 0);
    parallel_ranges(n, nt, [&](int t, int64_t b, int64_t e) {
        int64_t f = 0;
        for (int64_t i = b; i < e; i++)
        {
            const int64_t ri = inv ? inv[i] : row0 + i;
            for (int32_t p = rowptr[i]; p < rowptr[i + 1]; p++)
            {
                const int64_t cj = inv ? inv[colind[p]] : colind[p];
                f += (std::llabs(cj - ri) > window);
            }
        }
        far[size_t(t)] = f;
    });
    int64_t total_far = 0;
    for (int64_t f : far)
        total_far += f;
    const int64_t total = n > 0 ? int64_t(rowptr[n]) - int64_t(rowptr[0]) : 0;
    return total ? double(total_far) / double(total) : 0.0;
}

namespace {
// The give-up test of rcm_order evaluated BEFORE anything serial is built, for a structurally symmetric pattern: a level-
// synchronous breadth-first search from the vertex the ordering would start from (minimum degree, lowest index), the frontier
// expanded by the host threads straight from the CSR rows.  The set of vertices of a level does not depend on the order in
// which threads reach them, so the widths are those of the serial search.  Returns true (and the width) as soon as a level is
// wider than `fraction` of the whole matrix with at least 4096 vertices reached — then no ordering can localise the gathers and
// the seconds a graph copy, a degree sort and a full serial search of a 10M-row expander would cost are saved.
bool expander_precheck(int64_t n, const int32_t* rowptr, const int32_t* colind, double fraction, int64_t* widest_out)
{
    if (n < 65536 || fraction <= 0.0)
        return false;
    const int nt = int(std::max<int64_t>(1, std::min<int64_t>(ingest_threads(), n / 65536)));
    // start vertex: minimum degree without counting the diagonal, lowest index among equals (rcm_order's first seed)
    std::vector<std::pair<int32_t, int64_t>> best(static_cast<size_t>(nt), {INT32_MAX, -1});
    parallel_ranges(n, nt, [&](int t, int64_t b, int64_t e) {
        std::pair<int32_t, int64_t> m{INT32_MAX, -1};
        for (int64_t i = b; i < e; i++)
        {
            int32_t d = 0;
            for (int32_t p = rowptr[i]; p < rowptr[i + 1]; p++)
                d += (colind[p] != i);
            if (d < m.first)
                m = {d, i};
        }
        best[size_t(t)] = m;
    });
    std::pair<int32_t, int64_t> m{INT32_MAX, -1};
    for (const auto& b : best)
        if (b.second >= 0 && b.first < m.first)
            m = b;
    if (m.second < 0)
        return false;
    std::vector<uint8_t> seen(size_t(n), 0);
    std::vector<int32_t> frontier{int32_t(m.second)}, next;
    seen[size_t(m.second)] = 1;
    int64_t reached = 1, widest = 1;
    std::vector<std::vector<int32_t>> local(static_cast<size_t>(nt));
    while (!frontier.empty())
    {
        const int64_t fsz = int64_t(frontier.size());
        const int parts = int(std::max<int64_t>(1, std::min<int64_t>(nt, fsz / 256)));
        for (auto& l : local)
            l.clear();
        parallel_ranges(fsz, parts, [&](int t, int64_t b, int64_t e) {
            std::vector<int32_t>& out = local[size_t(t)];
            for (int64_t k = b; k < e; k++)
            {
                const int32_t v = frontier[size_t(k)];
                for (int32_t p = rowptr[v]; p < rowptr[v + 1]; p++)
                {
                    const int32_t u = colind[p];
                    if (u != v && !__atomic_load_n(&seen[size_t(u)], __ATOMIC_RELAXED) &&
                        __atomic_exchange_n(&seen[size_t(u)], uint8_t(1), __ATOMIC_RELAXED) == 0)
                        out.push_back(u);
                }
            }
        });
        next.clear();
        for (int t = 0; t < parts; t++)
            next.insert(next.end(), local[size_t(t)].begin(), local[size_t(t)].end());
        frontier.swap(next);
        const int64_t width = int64_t(frontier.size());
        reached += width;
        widest = std::max(widest, width);
        if (reached >= 4096 && double(widest) > fraction * double(n))
        {
            if (widest_out)
                *widest_out = widest;
            return true;
        }
    }
    return false;
}
}  // namespace

bool rcm_order(int64_t n, const int32_t* rowptr, const int32_t* colind, bool symmetric_pattern, double max_level_fraction,
               std::vector<int32_t>& perm, ReorderStats* stats)
{
    perm.clear();
    if (stats)
        *stats = ReorderStats{};
    if (n <= 0)
        return true;
    int64_t pre_widest = 0;
    if (symmetric_pattern && expander_precheck(n, rowptr, colind, max_level_fraction, &pre_widest))
    {
        if (stats)
        {
            stats->gave_up = true;
            stats->widest_level = pre_widest;
            stats->first_component = n;
        }
        return false;
    }
    const Graph g = build_graph(n, rowptr, colind, symmetric_pattern);
    std::vector<uint8_t> done(size_t(n), 0);
    Bfs s;
    s.order.resize(size_t(n));
    s.level.assign(size_t(n), -1);
    perm.reserve(size_t(n));
    // vertices by increasing degree: component seeds are minimum-degree vertices
    std::vector<int32_t> by_degree(static_cast<size_t>(n));
    std::iota(by_degree.begin(), by_degree.end(), 0);
    std::stable_sort(by_degree.begin(), by_degree.end(), [&](int32_t a, int32_t b) { return g.degree(a) < g.degree(b); });
    std::vector<int32_t> nb;
    int64_t components = 0, widest_all = 0;
    for (int64_t seed_i = 0; seed_i < n; seed_i++)
    {
        int32_t start = by_degree[size_t(seed_i)];
        if (done[size_t(start)])
            continue;
        components++;
        // pseudo-peripheral vertex
        s.count = 0;
        bfs(g, start, done, s);
        if (components == 1 && max_level_fraction > 0.0 && s.count >= 4096 && double(s.widest) > max_level_fraction * double(s.count))
        {
            if (stats)
            {
                stats->gave_up = true;
                stats->widest_level = s.widest;
                stats->first_component = s.count;
            }
            perm.clear();
            return false;  // expander-like: no ordering localises the gathers
        }
        for (int round = 0; round < 8; round++)
        {
            int32_t cand = s.order[size_t(s.last_begin)];
            for (int64_t i = s.last_begin; i < s.count; i++)
                if (g.degree(s.order[size_t(i)]) < g.degree(cand))
                    cand = s.order[size_t(i)];
            const int ecc = s.nlevels;
            const int64_t width = s.widest;
            bfs(g, cand, done, s);
            const bool better = s.nlevels > ecc || (s.nlevels == ecc && s.widest < width);
            start = cand;
            if (!better)
                break;
        }
        widest_all = std::max(widest_all, s.widest);
        // Cuthill-McKee numbering of this component from `start`
        const size_t base = perm.size();
        perm.push_back(start);
        done[size_t(start)] = 1;
        for (size_t head = base; head < perm.size(); head++)
        {
            const int32_t v = perm[head];
            nb.clear();
            for (int64_t p = g.ptr[size_t(v)]; p < g.ptr[size_t(v) + 1]; p++)
            {
                const int32_t u = g.adj[size_t(p)];
                if (!done[size_t(u)])
                {
                    done[size_t(u)] = 1;
                    nb.push_back(u);
                }
            }
            std::sort(nb.begin(), nb.end(), [&](int32_t a, int32_t b) {
                const int32_t da = g.degree(a), db = g.degree(b);
                return da != db ? da < db : a < b;
            });
            perm.insert(perm.end(), nb.begin(), nb.end());
        }
        // level[] entries of this component stay >= 0, which keeps them out of later searches together with done[]
        s.count = 0;
    }
    std::reverse(perm.begin(), perm.end());
    if (stats)
    {
        stats->components = components;
        stats->widest_level = widest_all;
    }
    return true;
}

void permute_csr(int64_t n, const int32_t* rowptr, const int32_t* colind, const double* val, const std::vector<int32_t>& perm,
                 std::vector<int32_t>& rp, std::vector<int32_t>& ci, std::vector<double>& v)
{
    std::vector<int32_t> inv(static_cast<size_t>(n));
    for (int64_t i = 0; i < n; i++)
        inv[size_t(perm[size_t(i)])] = int32_t(i);
    rp.assign(size_t(n) + 1, 0);
    for (int64_t i = 0; i < n; i++)
    {
        const int32_t o = perm[size_t(i)];
        rp[size_t(i) + 1] = rp[size_t(i)] + (rowptr[o + 1] - rowptr[o]);
    }
    ci.resize(size_t(rp[size_t(n)]));
    v.resize(size_t(rp[size_t(n)]));
    const int nt = int(std::max<int64_t>(1, std::min<int64_t>(ingest_threads(), n / 16384)));
    parallel_ranges(n, nt, [&](int, int64_t b, int64_t e) {
        std::vector<std::pair<int32_t, double>> row;
        for (int64_t i = b; i < e; i++)
        {
            const int32_t o = perm[size_t(i)];
            row.clear();
            for (int32_t p = rowptr[o]; p < rowptr[o + 1]; p++)
                row.emplace_back(inv[size_t(colind[p])], val[p]);
            // ascending new column; equal columns (duplicates) keep their storage order
            std::stable_sort(row.begin(), row.end(),
                             [](const std::pair<int32_t, double>& a, const std::pair<int32_t, double>& c) { return a.first < c.first; });
            int32_t q = rp[size_t(i)];
            for (const auto& en : row)
            {
                ci[size_t(q)] = en.first;
                v[size_t(q)] = en.second;
                q++;
            }
        }
    });
}

}  // namespace mispec

// Host-only entry point (no device needed): the ordering itself, for tests and for callers that want to apply it
// themselves.  perm_out[new] = old.  Returns MISPEC_OK; *gave_up = 1 (perm_out = identity) when the pattern is
// expander-like and no ordering was produced.
extern "C" int mispec_rcm_order(int64_t n, const int32_t* rowptr, const int32_t* colind, int symmetric_pattern, int32_t* perm_out,
                                int* gave_up, int64_t* widest_level)
{
    return mispec::guarded([&] {
        MISPEC_REQUIRE(n >= 0 && rowptr && perm_out && (colind || rowptr[n] == 0), "mispec_rcm_order: bad argument");
        for (int64_t i = 0; i < n; i++)
            for (int32_t p = rowptr[i]; p < rowptr[i + 1]; p++)
                MISPEC_REQUIRE(colind[p] >= 0 && colind[p] < n, "mispec_rcm_order: column index out of range");
        std::vector<int32_t> perm;
        mispec::ReorderStats st;
        const bool ok = mispec::rcm_order(n, rowptr, colind, symmetric_pattern != 0, 0.125, perm, &st);
        if (ok)
            std::memcpy(perm_out, perm.data(), size_t(n) * sizeof(int32_t));
        else
            for (int64_t i = 0; i < n; i++)
                perm_out[i] = int32_t(i);
        if (gave_up)
            *gave_up = ok ? 0 : 1;
        if (widest_level)
            *widest_level = st.widest_level;
    });
}
