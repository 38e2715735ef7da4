// mc_oracle_seed.cpp -- CPU ORACLE, seed mode.  TEST INFRASTRUCTURE ONLY (see mc_oracle.h).
//
// Restates the seed branch of Marching::recalculate (Source/marching.cpp:310-331) with
// find_cubes_for_seeding (:42-101), get_starting_seed_grid (:104-113) and the xyz tolerance
// comparator (Source/marching.h:32-55), using the same containers the reference uses (std::deque,
// std::set<xyz>), so that revisits caused by float drift of the cell positions would show here as
// they do there.  The per-cell work (calculate_step + add_step_to_poly_data) is mc_oracle.c's.
#include <cmath>
#include <cstdint>
#include <deque>
#include <set>

#include "../include/mc_tables_data.h"
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

const unsigned short k_face_corner[6] = MC_FACE_CORNER_INIT;  // marching_lookup.h:25-32 cube_face_vertex_table
// marching_lookup.h:43-50 cube_face_normal: face i is the one whose 4 corners share a coordinate; corner v has
// x = bit of 0x66, y = bit of 0xCC, z = v >> 2 (marching.cpp:471-472)
void face_normal(int f, int n[3]) {
    int bx = 0, by = 0, bz = 0;
    for (int i = 0; i < 4; ++i) {
        const int v = (k_face_corner[f] >> (4 * i)) & 0xF;
        bx += (0x66 >> v) & 1;
        by += (0xCC >> v) & 1;
        bz += v >> 2;
    }
    n[0] = bx == 4 ? 1 : bx == 0 ? -1 : 0;  // the other two axes have two corners on each side (sum 2)
    n[1] = by == 4 ? 1 : by == 0 ? -1 : 0;
    n[2] = bz == 4 ? 1 : bz == 0 ? -1 : 0;
}

}  // namespace

extern "C" int orc_march_seed(const char* eq, float step, float iso, const float scale[3], int pow_mode, int want,
                              const float seed[3], orc_mesh* out) {
    if (!((double)step >= 0.001 && (double)step <= .5)) return -3;
    for (int a = 0; a < 3; ++a)
        if (!(seed[a] <= 1 && seed[a] >= -1)) return -5;  // marching.cpp:128
    void* h = orc_seed_begin(eq, step, iso, scale, pow_mode, want);
    if (!h) return -1;
    std::deque<xyz> seed_queue;
    std::set<xyz> my_seed_set;
    // :104-113 get_starting_seed_grid
    const float dx = ((seed[0] / scale[0] - (-1)) / step);
    const float dy = ((seed[1] / scale[1] - (-1)) / step);
    const float dz = ((seed[2] / scale[2] - (-1)) / step);
    xyz g;
    g.x = -1 + std::floor(dx) * step;
    g.y = -1 + std::floor(dy) * step;
    g.z = -1 + std::floor(dz) * step;
    g.idx = -1;
    seed_queue.push_back(g);  // :319-320
    my_seed_set.insert(g);
    uint64_t visited = 0;
    int err = 0;
    while (!seed_queue.empty()) {  // :324-330
        const xyz cur = seed_queue.front();
        seed_queue.pop_front();
        uint8_t code = 0;
        const int r = orc_seed_cell(h, cur.x, cur.y, cur.z, &code);  // calculate_step + add_step_to_poly_data
        if (r) { err = r == -4 ? -4 : -2; break; }
        ++visited;
        // :42-101 find_cubes_for_seeding.  edge_list is empty unless the cell produced a surface (:508-510 returns
        // before it is filled); an edge has an intersection when its two corners differ (:566).
        if (code == 0 || code == 255) continue;
        for (int f = 0; f < 6; ++f) {
            bool any = false, first = (code >> (k_face_corner[f] & 0xF)) & 1;
            for (int i = 1; i < 4; ++i)
                if ((((code >> ((k_face_corner[f] >> (4 * i)) & 0xF)) & 1) != 0) != first) any = true;
            if (!any) continue;  // :62-69: no edge of this face carries an intersection
            int n[3];
            face_normal(f, n);
            xyz nx;
            nx.x = cur.x + n[0] * step;  // :78-80
            nx.y = cur.y + n[1] * step;
            nx.z = cur.z + n[2] * step;
            nx.idx = -1;
            const double hs = 0.5 * step;
            if (nx.x >= -1 - hs && (nx.x + hs) <= 1 && nx.y >= -1 - hs && (nx.y + hs) <= 1 && nx.z >= -1 - hs &&
                (nx.z + hs) <= 1) {                            // :84-86
                if (my_seed_set.insert(nx).second) seed_queue.push_back(nx);  // :89-97
            }
        }
    }
    orc_seed_finish(h, out, visited);
    if (err) {
        orc_mesh_free(out);
        return err;
    }
    return 0;
}
