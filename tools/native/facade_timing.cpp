// facade_timing.cpp -- what a reference-style caller waits for: wall time of Marching::recalculate() (sweep + welded mesh on
// the GPU + the copy into Poly_Data's std::vectors) at a given grid_res, five calls in a row.
// Build: g++ -std=c++14 -O2 -Iinclude/compat tools/native/facade_timing.cpp -Lmarching-cube-for-implicit-surfaces_amd -lmc_hip ...
#include "marching.h"
#include "evaluator.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>

int main(int argc, char** argv) {
    const int grid_res = argc > 1 ? std::atoi(argv[1]) : 1024;
    const bool indexed = argc > 2 ? std::atoi(argv[2]) != 0 : true;
    Evaluator evaluator;
    evaluator.set_equation("x^2+y^2+z^2-1");
    Marching m;
    m.set_evaluator(&evaluator);
    m.set_grid_step_size(2.0f / (float)grid_res);
    m.set_indexed(indexed);
    for (int i = 0; i < 6; ++i) {
        const auto t0 = std::chrono::steady_clock::now();
        if (!m.recalculate()) {
            std::printf("error: %s\n", m.last_error().c_str());
            return 1;
        }
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        const mc_result& r = m.last_result();
        std::printf("recalculate %d: %.2f ms wall (GPU: sweep %.3f + index %.3f ms; %zu vertices, %zu triangles, interpreted %d)\n", i, ms,
                    (double)r.ms_total, (double)r.ms_index, m.get_poly_data()->vertex_list.size() / 3, m.get_poly_data()->tri_list.size() / 3, r.interpreted);
    }
    return 0;
}
