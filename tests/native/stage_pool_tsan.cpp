// Back-to-back jobs on the staging pool with more threads than cores, under ThreadSanitizer (tests/test_stage_pool.py builds
// this with g++ -fsanitize=thread against pc-accumulation-lib_amd/csrc/pca_stage_pool.h).  Every job copies a fresh pattern
// through a slice table that is FREED right after run() returns -- a helper holding a claim of the previous job would read
// freed memory or copy a slice twice -- and checks the destination byte for byte.
#include <stdio.h>
#include <memory>
#include "pca_stage_pool.h"

int main(int argc, char **argv)
{
    const int helpers = argc > 1 ? atoi(argv[1]) : 24, jobs = argc > 2 ? atoi(argv[2]) : 3000;
    setenv("PCA_STAGING_SPIN_US", "50", 1);                 // helpers also go through the sleep / wake path
    pca_stage::Pool pool;
    pool.start(helpers);
    constexpr size_t SLICE = 4096;
    std::vector<char> src(SLICE * 64), dst(SLICE * 64);
    long bad = 0;
    for (int j = 0; j < jobs; ++j) {
        const int n = 1 + (j * 7) % 64;                     // slice counts go up and down: an index beyond the old total is inside the new one
        for (int i = 0; i < n; ++i) memset(src.data() + SLICE * i, (j + 3 * i) & 0xff, SLICE);
        memset(dst.data(), 0xee, dst.size());
        {
            std::unique_ptr<pca_stage::Slice[]> tab(new pca_stage::Slice[n]);
            for (int i = 0; i < n; ++i) tab[i] = {dst.data() + SLICE * i, src.data() + SLICE * i, SLICE - (size_t)(i % 5)};
            pool.run(tab.get(), n);
        }
        for (int i = 0; i < n; ++i) {
            const size_t len = SLICE - (size_t)(i % 5);
            if (memcmp(dst.data() + SLICE * i, src.data() + SLICE * i, len) != 0) ++bad;
            if (i % 5 && (unsigned char)dst[SLICE * i + len] != 0xee) ++bad;
        }
        for (size_t k = SLICE * n; k < dst.size(); k += 997) if ((unsigned char)dst[k] != 0xee) ++bad;
        if (j % 500 == 0) std::this_thread::sleep_for(std::chrono::microseconds(300));   // let them fall asleep now and then
    }
    printf("stage_pool: %d jobs, %d helpers, %ld bad\n", jobs, helpers, bad);
    return bad ? 1 : 0;
}
