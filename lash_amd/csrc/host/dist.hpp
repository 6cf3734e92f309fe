// dist.hpp — `lash dist` (/root/reference/src/main.rs:280-617, utils.rs:84-373): the consumer of the sketch files.
// SURVEY.md §8(f) row f2 ("next"): not part of the round-1 hot path.
#pragma once
#include <string>
#include <vector>

#include "../../../include/lash_gfx950.h"

namespace lashhost {

struct DistOptions {
    std::string query_prefix, ref_prefix, output_file = "dist", estimator = "fgra";
    int model = 1;          // 1 = Poisson: min(-ln(f)/k, 1); 0 = binomial: 1 - f^(1/k)   (main.rs:415-423)
    int threads = 1;
    bool fp32 = false, matrix = false;
    bool file_order = false;   // rows / columns / triangle in list-file order instead of the reference's map order (name_order.hpp)
    int device = 0;
    std::vector<int> devices;  // --devices 0,1,...: one worker per entry, blocks of reference rows in turn; empty = {device}
    uint32_t block_rows = 0;   // reference rows per GPU call; 0 = as many as keep the pair tables under ~0.5 GB
    std::string hll_bias_file; // --hll-bias / $LASH_HLL_BIAS: HLL++ bias tables (lash_hll_bias_load); empty = that regime is refused
    lash_layout layout;        // --layout / $LASH_LAYOUT (include/lash_gfx950.h)
    DistOptions() { lash_layout_default(&layout); }
};

std::string run_dist(const DistOptions &opt);

}  // namespace lashhost
