// reference_caller.cpp -- a caller written the way the reference's own callers are: Source/main.cpp:11-14 and the
// usage block of Source/marching_test_drawer.h:7-15, with the two #include lines exactly as they are there.  Built with
// -Iinclude/compat it gets the GPU-backed classes; nothing else differs.  Prints what tests/test_facade.py compares
// with the oracle's replay of the reference's std::set welding (and with SURVEY section 4's counts).
#include "marching.h"
#include "evaluator.h"

#include <cstdint>
#include <cstdio>

static uint64_t fnv1a(const void* p, size_t n) {
    uint64_t h = 1469598103934665603ull;
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; ++i) {
        h ^= b[i];
        h *= 1099511628211ull;
    }
    return h;
}

int main(int argc, char** argv) {
    Evaluator evaluator;  // evaluates equations (default "x+y")
    if (evaluator.evaluate(1.0f, 2.0f, 3.0f) != 3.0f) return 2;
    evaluator.set_equation(argc > 1 ? argv[1] : "x^2+y^2+z^2-1");

    Marching march_maker;  // generate the implicit surface mesh
    march_maker.set_evaluator(&evaluator);
    march_maker.set_grid_step_size(2.0f / 32);
    march_maker.set_surface_constant(0.0f);
    if (!march_maker.recalculate()) return 3;

    const Poly_Data* pData = march_maker.get_poly_data();
    std::vector<float> vNormal = CalculateNormal(pData);  // what the drawer does with it (drawer.cpp:795-801)
    printf("verts=%zu tris=%zu fnv_vertex_list=%016llx fnv_tri_list=%016llx normals=%zu f(0.5,0.5,0.5)=%.9g\n", pData->vertex_list.size() / 3,
           pData->tri_list.size() / 3, (unsigned long long)fnv1a(pData->vertex_list.data(), pData->vertex_list.size() * 4),
           (unsigned long long)fnv1a(pData->tri_list.data(), pData->tri_list.size() * 4), vNormal.size() / 3,
           (double)evaluator.evaluate(0.5f, 0.5f, 0.5f));
    return 0;
}
