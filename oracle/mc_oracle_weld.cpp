// mc_oracle_weld.cpp -- CPU ORACLE, indexed mesh.  TEST INFRASTRUCTURE ONLY (see mc_oracle.h).
//
// Restates how the reference turns the per-cell results of its sweep into Poly_Data:
//   Marching::add_step_to_poly_data / add_point / add_triangle  (Source/marching.cpp:599-654)
//   struct xyz and its tolerance comparator                     (Source/marching.h:32-55)
//   CalculateNormal                                             (Source/normal.h:3-41; glm cross / normalize)
// with the same container (std::set<xyz>) fed in the same order -- cells in sweep order (marching.cpp:372-383), inside
// a cell the crossed edges in edge order (:557-583) -- so that whatever the not-quite-an-ordering comparator makes the
// set do there it does here.  The per-cell work (calculate_step) is mc_oracle.c's.
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>

#include "mc_oracle.h"

namespace {

struct xyz {  // marching.h:32-55
    float x, y, z;
    int idx;
    static bool close_enough(float a, float b) { return std::abs(a - b) < 0.000001; }
    bool operator<(const xyz& r) const {
        if (!close_enough(x, r.x)) return x < r.x;
        if (!close_enough(y, r.y)) return y < r.y;
        if (!close_enough(z, r.z)) return z < r.z;
        return false;
    }
};

}  // namespace

extern "C" void orc_indexed_free(orc_indexed* m) {
    if (!m) return;
    free(m->vertex_list);
    free(m->tri_list);
    free(m->normals);
    m->vertex_list = m->normals = nullptr;
    m->tri_list = nullptr;
}

extern "C" int orc_march_indexed(const char* eq, float step, float iso, const float scale[3], int pow_mode,
                                 const orc_constraint* cons, int ncons, int z_begin, int z_end, orc_indexed* out) {
    memset(out, 0, sizeof(*out));
    if (!((double)step >= 0.001 && (double)step <= .5)) return -3;
    void* h = orc_step_begin(eq, step, iso, scale, pow_mode, cons, ncons);
    if (!h) return -1;
    const int n1 = orc_step_n1(h);
    if (z_end < 0 || z_end > n1) z_end = n1;
    if (z_begin < 0) z_begin = 0;
    std::vector<float> vertex_list;
    std::vector<unsigned int> tri_list;
    std::set<xyz> vertex_set;
    int err = 0;
    for (int iz = z_begin; iz < z_end && !err; ++iz)
        for (int iy = 0; iy < n1 && !err; ++iy)
            for (int ix = 0; ix < n1; ++ix) {
                orc_step s;
                if (orc_step_cell(h, ix, iy, iz, &s)) { err = -2; break; }
                if (s.skipped || s.n_points == 0) continue;
                int v_i_list[12];  // marching.cpp:602-604
                for (int i = 0; i < 12; ++i) v_i_list[i] = -1;
                for (int i = 0; i < s.n_points; ++i) {  // :607-615
                    const float x = s.point[i][0], y = s.point[i][1], z = s.point[i][2];
                    if (!std::isnan(x)) {  // :627-643 add_point
                        const int new_vertex_i = (int)(vertex_list.size() / 3);
                        const int found = vertex_set.insert(xyz{x, y, z, new_vertex_i}).first->idx;
                        if (found == new_vertex_i) {
                            vertex_list.push_back(x);
                            vertex_list.push_back(y);
                            vertex_list.push_back(z);
                        }
                        v_i_list[i] = found;
                    }
                }
                for (int i = 0; i < 3 * s.n_tris; i += 3) {  // :618-624, :646-654
                    tri_list.push_back((unsigned int)v_i_list[s.tri_vlist[i]]);
                    tri_list.push_back((unsigned int)v_i_list[s.tri_vlist[i + 1]]);
                    tri_list.push_back((unsigned int)v_i_list[s.tri_vlist[i + 2]]);
                }
            }
    orc_step_end(h);
    if (err) return err;

    // normal.h:3-41 with glm's float operations spelled out: cross(x, y) = (x.y*y.z - y.y*x.z, x.z*y.x - y.z*x.x,
    // x.x*y.y - y.x*x.y); normalize(v) = v * inversesqrt(dot(v, v)) with inversesqrt = 1 / sqrt, dot = (x*x + y*y) + z*z
    const size_t nv = vertex_list.size() / 3, nt = tri_list.size() / 3;
    std::vector<float> n(3 * nv, 0.0f);
    const float* v = vertex_list.data();
    for (size_t i = 0; i < nt; ++i) {
        const unsigned i1 = tri_list[3 * i], i2 = tri_list[3 * i + 1], i3 = tri_list[3 * i + 2];
        if (i1 >= nv || i2 >= nv || i3 >= nv) continue;  // index -1 of a NaN point: the reference would read out of bounds
        volatile float bax = v[3 * i2] - v[3 * i1], bay = v[3 * i2 + 1] - v[3 * i1 + 1], baz = v[3 * i2 + 2] - v[3 * i1 + 2];
        volatile float cax = v[3 * i3] - v[3 * i1], cay = v[3 * i3 + 1] - v[3 * i1 + 1], caz = v[3 * i3 + 2] - v[3 * i1 + 2];
        volatile float p0 = bay * caz, p1 = cay * baz, p2 = baz * cax, p3 = caz * bax, p4 = bax * cay, p5 = cax * bay;
        const float nx = p0 - p1, ny = p2 - p3, nz = p4 - p5;
        const unsigned idx[3] = {i1, i2, i3};
        for (unsigned k : idx) {
            volatile float a = nx + n[3 * k], b = ny + n[3 * k + 1], c = nz + n[3 * k + 2];
            n[3 * k] = a;
            n[3 * k + 1] = b;
            n[3 * k + 2] = c;
        }
    }
    for (size_t i = 0; i + 2 < n.size(); i += 3) {
        volatile float xx = n[i] * n[i], yy = n[i + 1] * n[i + 1], zz = n[i + 2] * n[i + 2];
        volatile float d = xx + yy;
        d = d + zz;
        volatile float inv = 1.0f / std::sqrt((float)d);
        n[i] = n[i] * inv;
        n[i + 1] = n[i + 1] * inv;
        n[i + 2] = n[i + 2] * inv;
    }

    out->n_verts = nv;
    out->n_tris = nt;
    out->vertex_list = (float*)malloc((nv ? nv : 1) * 12);
    out->tri_list = (uint32_t*)malloc((nt ? nt : 1) * 12);
    out->normals = (float*)malloc((nv ? nv : 1) * 12);
    if (!out->vertex_list || !out->tri_list || !out->normals) {
        orc_indexed_free(out);
        return -4;
    }
    memcpy(out->vertex_list, vertex_list.data(), nv * 12);
    memcpy(out->tri_list, tri_list.data(), nt * 12);
    memcpy(out->normals, n.data(), nv * 12);
    return 0;
}
