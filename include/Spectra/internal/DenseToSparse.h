// Helper of the dense shift-solve / Cholesky operator classes: the compressed-column form of a dense host matrix, which is
// what the device factorisations ingest.  Exact zeros are not stored, so a dense matrix that happens to be banded still
// takes the banded device path.
#ifndef MISPEC_SPECTRA_DENSE_TO_SPARSE_H
#define MISPEC_SPECTRA_DENSE_TO_SPARSE_H

#include <vector>

#include "Dense.h"

namespace Spectra {
namespace internal {

struct CompressedCopy
{
    std::vector<int> outer, inner;
    std::vector<double> values;
    Index rows = 0, cols = 0;

    explicit CompressedCopy(const DenseView<double>& A) : rows(A.rows), cols(A.cols)
    {
        outer.reserve(static_cast<std::size_t>(A.cols) + 1);
        outer.push_back(0);
        for (Index j = 0; j < A.cols; j++)
        {
            for (Index i = 0; i < A.rows; i++)
            {
                const double v = A.row_major ? A.data[i * A.ld + j] : A.data[j * A.ld + i];
                if (v != 0.0)
                {
                    inner.push_back(static_cast<int>(i));
                    values.push_back(v);
                }
            }
            outer.push_back(static_cast<int>(inner.size()));
        }
    }
    SparseView<double, int> view() const
    {
        SparseView<double, int> v;
        v.rows = rows;
        v.cols = cols;
        v.outer = outer.data();
        v.inner = inner.data();
        v.values = values.data();
        v.row_major = false;
        return v;
    }
};

}  // namespace internal
}  // namespace Spectra

#endif
