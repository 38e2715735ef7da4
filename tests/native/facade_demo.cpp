// facade_demo.cpp -- the reference's documented usage (Source/marching_test_drawer.h:7-15,
// Source/main.cpp:11-14) written against include/mc_marching.hpp.  Prints counts and the FNV-1a
// fingerprint of the triangle soup so tests/test_facade.py can compare with the fingerprints
// SURVEY.md section 4 recorded from the unmodified reference.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>

#include "mc_marching.hpp"

using namespace mc_amd;

int main(int argc, char** argv) {
    const char* eq = argc > 1 ? argv[1] : "x^2+y^2+z^2-1";
    const int grid_res = argc > 2 ? atoi(argv[2]) : 32;
    const float iso = argc > 3 ? (float)atof(argv[3]) : 0.0f;
    const bool indexed = argc > 4 && strcmp(argv[4], "indexed") == 0;
    const float scale = argc > 5 ? (float)atof(argv[5]) : 1.0f;
    const float step_override = argc > 6 ? (float)atof(argv[6]) : 0.0f;
    const char* ply = argc > 7 ? argv[7] : nullptr;
    try {
        Context ctx(0);
        Evaluator evaluator;                 // default equation "x+y" (evaluator.cpp:6-8)
        if (!evaluator.set_equation(eq)) {
            printf("parse_error\n");
            return 2;
        }
        if (evaluator.set_equation("sin(x)")) return 3;      // rejected: the previous equation stays
        Marching march_maker(ctx);
        march_maker.set_evaluator(&evaluator);
        if (march_maker.set_grid_step_size(0.6f)) return 4;  // outside [0.001, 0.5]
        march_maker.set_grid_step_size(step_override > 0.0f ? step_override : 2.0f / (float)grid_res);
        march_maker.set_surface_constant(iso);
        march_maker.set_scaling_x(scale);
        march_maker.set_scaling_y(scale);
        march_maker.set_scaling_z(scale);
        march_maker.set_indexed(indexed);
        if (const char* sd = getenv("MC_DEMO_SEED")) {  // "x y z": seed mode (marching.cpp:115-137)
            float x, y, z;
            if (sscanf(sd, "%f %f %f", &x, &y, &z) != 3) return 13;
            if (march_maker.set_seed(2.0f, 0.0f, 0.0f)) return 14;  // outside [-1,1]: refused, the old seed stays
            if (!march_maker.set_seed(x, y, z)) return 15;
            march_maker.seed_mode(true);
        }
        if (const char* c = getenv("MC_DEMO_CONSTRAINT")) {  // "lhs op rhs", the developer viewer's `x > -0.5` hotkey
            char lhs[128], op[8];
            float rhs;
            if (sscanf(c, "%127s %7s %f", lhs, op, &rhs) != 3) return 8;
            if (march_maker.set_constraint0(lhs, "=>", rhs)) return 9;  // not one of the four spellings
            if (!march_maker.set_constraint0(lhs, op, rhs) || !march_maker.use_constraint0(true)) return 10;
        }
        if (!march_maker.recalculate()) {
            printf("error: %s\n", march_maker.last_error().c_str());
            return 5;
        }
        const Poly_Data* pd = march_maker.get_poly_data();
        uint64_t h = 1469598103934665603ull;
        const unsigned char* b = reinterpret_cast<const unsigned char*>(pd->vertex_list.data());
        for (size_t i = 0; i < pd->vertex_list.size() * sizeof(float); ++i) {
            h ^= b[i];
            h *= 1099511628211ull;
        }
        printf("cells_per_axis=%d tris=%zu verts=%zu fnv_soup=%016llx f(1,2,3)=%g\n", march_maker.last_result().cells_per_axis,
               pd->tri_list.size() / 3, pd->vertex_list.size() / 3, (unsigned long long)h, evaluator.evaluate(ctx, 1, 2, 3));
        if (getenv("MC_DEMO_STEPS")) {  // the one-cell teaching trace, for EVERY cell in sweep order: its triangles
            const int n1 = march_maker.last_result().cells_per_axis;   // concatenated must be the sweep's soup
            uint64_t hs = 1469598103934665603ull, hc = 1469598103934665603ull;
            size_t ntri = 0;
            Step_Data sd;
            for (int iz = 0; iz < n1; ++iz)
                for (int iy = 0; iy < n1; ++iy)
                    for (int ix = 0; ix < n1; ++ix) {
                        if (!march_maker.step_at(ix, iy, iz, &sd)) return 12;
                        const unsigned char cb = (unsigned char)sd.cube_code;
                        hc = (hc ^ cb) * 1099511628211ull;
                        for (int v : sd.tri_vlist) {
                            const unsigned char* q = reinterpret_cast<const unsigned char*>(&sd.intersect_coord[3 * (size_t)v]);
                            for (int i = 0; i < 12; ++i) hs = (hs ^ q[i]) * 1099511628211ull;
                        }
                        ntri += sd.tri_vlist.size() / 3;
                    }
            printf("steps: tris=%zu fnv_codes=%016llx fnv_soup=%016llx\n", ntri, (unsigned long long)hc, (unsigned long long)hs);
        }
        if (const char* nf = getenv("MC_DEMO_NORMALS")) {  // dump normal_list (indexed mode: CalculateNormal of normal.h)
            FILE* f = fopen(nf, "w");
            if (!f) return 11;
            for (size_t i = 0; i + 2 < pd->normal_list.size(); i += 3)
                fprintf(f, "%.9g %.9g %.9g\n", pd->normal_list[i], pd->normal_list[i + 1], pd->normal_list[i + 2]);
            fclose(f);
        }
        if (getenv("MC_DEMO_TWO_OBJECTS")) {
            // constraints, seed mode and the seed belong to ONE Marching object (marching.h:58-69, :130-157), also when
            // several objects share the process-wide context of the argument-less constructors
            Evaluator ev2;
            if (!ev2.set_equation(eq)) return 20;
            size_t plain = 0, cons = 0;
            {
                Marching m1, m2;  // both on default_context()
                for (Marching* m : {&m1, &m2}) {
                    m->set_evaluator(&ev2);
                    m->set_grid_step_size(2.0f / (float)grid_res);
                    m->set_surface_constant(iso);
                }
                if (!m1.set_constraint0("x", ">", -0.5f) || !m1.use_constraint0(true)) return 21;
                if (!m2.recalculate()) return 22;  // m1's constraint must not apply here
                plain = m2.get_poly_data()->tri_list.size() / 3;
                if (!m1.recalculate()) return 23;
                cons = m1.get_poly_data()->tri_list.size() / 3;
                if (!m2.recalculate() || m2.get_poly_data()->tri_list.size() / 3 != plain) return 24;
                m1.set_seed(0.0f, 0.0f, 1.0f);
                m1.seed_mode(true);                // m2 stays a dense, indexed sweep
                if (!m2.recalculate() || m2.get_poly_data()->tri_list.size() / 3 != plain) return 25;
                if (m2.get_poly_data()->vertex_list.size() / 3 >= 3 * plain) return 26;  // welded, not soup
                if (!m1.recalculate()) return 27;
                // step_at honours the object's own constraint and lands in poly_data.step_data (marching.h:29)
                Step_Data sd;
                if (!m1.step_at(0, grid_res / 2, grid_res / 2, &sd) || !sd.tri_vlist.empty()) return 28;  // x = -1: outside x > -0.5
                if (m1.get_poly_data()->step_data.step_i != sd.step_i || sd.step_i < 1) return 29;
                if (!m2.step_at(0, grid_res / 2, grid_res / 2, &sd)) return 30;
                if (m2.get_poly_data()->step_data.corner_values.size() != 8) return 31;
            }
            Marching m3;  // a destroyed object leaves nothing behind in the shared context
            m3.set_evaluator(&ev2);
            m3.set_grid_step_size(2.0f / (float)grid_res);
            m3.set_surface_constant(iso);
            if (!m3.recalculate() || m3.get_poly_data()->tri_list.size() / 3 != plain) return 32;
            printf("two_objects=ok plain=%zu constrained=%zu\n", plain, cons);
        }
        if (ply) {  // PLY round trip through the reference's on-disk format (marching.cpp:665-854)
            if (!march_maker.save_poly_to_file(ply)) return 6;
            const size_t nv = pd->vertex_list.size(), ni = pd->tri_list.size();
            if (!march_maker.load_poly_from_file(ply)) return 7;   // appends, like the reference
            const Poly_Data* q = march_maker.get_poly_data();
            bool same = q->vertex_list.size() == 2 * nv && q->tri_list.size() == 2 * ni;
            for (size_t i = 0; same && i < ni; ++i) same = q->tri_list[i] == q->tri_list[ni + i];
            float maxd = 0.0f;
            for (size_t i = 0; same && i < nv; ++i) maxd = std::fmax(maxd, std::fabs(q->vertex_list[i] - q->vertex_list[nv + i]));
            printf("ply_roundtrip=%s maxd=%g\n", same ? "ok" : "MISMATCH", maxd);
        }
    } catch (const std::exception& e) {
        printf("exception: %s\n", e.what());
        return 1;
    }
    return 0;
}
