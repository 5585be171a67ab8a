// Host-side symmetric reordering of a sparse pattern (reverse Cuthill-McKee), see reorder.hip.
#pragma once
#include <vector>

#include "common.hpp"

namespace mispec {

struct ReorderStats
{
    bool gave_up = false;        // expander-like pattern: the first BFS level structure was too wide, no ordering produced
    int64_t widest_level = 0;    // widest BFS level seen (~ the bandwidth after reordering)
    int64_t first_component = 0;
    int64_t components = 0;
};

// Fraction of the stored entries whose column is more than `window` positions away from their row; with inv != nullptr
// rows and columns are first mapped through inv (old -> new), i.e. the measure of the reordered matrix.
// rows [row0, row0 + n) of the matrix the arrays describe (a row shard: inv must then be nullptr)
double far_fraction(int64_t n, const int32_t* rowptr, const int32_t* colind, const int32_t* inv, int64_t window, int64_t row0 = 0);

// perm[new] = old.  symmetric_pattern: the pattern is known to be structurally symmetric (else A + A' is used).
// max_level_fraction > 0: give up (return false, perm empty) when the widest level of the first BFS exceeds that
// fraction of its component (>= 4096 vertices).
bool rcm_order(int64_t n, const int32_t* rowptr, const int32_t* colind, bool symmetric_pattern, double max_level_fraction,
               std::vector<int32_t>& perm, ReorderStats* stats);

// B = P A P' with B(i, j) = A(perm[i], perm[j]); rows of B sorted by column.
void permute_csr(int64_t n, const int32_t* rowptr, const int32_t* colind, const double* val, const std::vector<int32_t>& perm,
                 std::vector<int32_t>& rp, std::vector<int32_t>& ci, std::vector<double>& v);

}  // namespace mispec
