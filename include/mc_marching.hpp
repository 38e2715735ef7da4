// mc_marching.hpp -- header-only C++ facade over the C ABI (include/mc_hip.h) that mirrors the
// reference's operator surface for the hot path, so a caller written against
//   class Evaluator  (Source/evaluator.h:24-86)   and
//   class Marching   (Source/marching.h:72-157)
// switches by changing the include and linking libmc_hip.so.  Same method names, argument
// meaning and error behaviour; the sweep itself runs on the GPU.
//
//   reference                                   this facade
//   ------------------------------------------  -------------------------------------------------
//   Evaluator::set_equation(string) -> bool      same (tokenizer accept/reject, evaluator.cpp:15)
//   Evaluator(string) throws on parse error      same (evaluator.cpp:10-13)
//   Evaluator::evaluate(x,y,z) -> float          same value (P1 power rule), computed on the GPU
//   Marching::set_evaluator(Evaluator*) -> bool  same (marching.cpp:140-147), pointer is borrowed
//   Marching::set_grid_step_size(float) -> bool  same range check [0.001, 0.5] (marching.cpp:226)
//   Marching::set_surface_constant(float)        same (marching.cpp:149)
//   Marching::set_scaling_{x,y,z}(float)         same (marching.cpp:240-251)
//   Marching::recalculate() -> bool              full sweep (marching.cpp:368-384) on the GPU
//   Marching::get_poly_data() -> Poly_Data*      same layout: vertex_list float xyz, tri_list u32
//
//   Marching::save_poly_to_file / load_poly_from_file   same ASCII PLY (marching.cpp:665-854), path argument
//                                                instead of the Win32 dialog
//
// Differences, all documented in DESIGN.md: by default the mesh is the triangle SOUP the GPU emits
// (tri_list = 0..3T-1) with `normal_list` (gradient normals, 3 floats per vertex) as an extra
// member; set_indexed(true) additionally runs the reference's vertex welding on the host -- the
// same std::set<xyz> with the same tolerance comparator fed in the same order
// (marching.cpp:599-643, marching.h:32-55), so vertex_list / tri_list come out as the reference
// builds them; normal_list then holds the reference's own area-weighted vertex normals (CalculateNormal,
// Source/normal.h:3-41, also available as a free function).  Constraints (set_constraint0..2 / use_constraint0..2)
// and seed mode (seed_mode / set_seed) are provided -- seed mode returns the same triangles as the reference's
// walk, in sweep order (mc_hip.h).  The step-by-step STATE MACHINE of recalculate() is not; step_at(ix, iy, iz)
// gives the Step_Data of any one cell instead.  A failed GPU call makes recalculate() return false and
// last_error() non-empty instead of crashing.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <exception>
#include <set>
#include <stdexcept>
#include <string>
#include <vector>

#include "mc_hip.h"
#include "mc_tables_data.h"

namespace mc_amd {

// marching.h:15-23: what the reference records while it works on ONE grid cell (its step-by-step teaching view).
// The sweep never builds it; Marching::step_at(ix, iy, iz) fills one for a chosen cell.
struct Step_Data {
    int step_i = -2;
    std::vector<float> corner_coords;    // coordinates of the 8 corners, size 24 (marching.cpp:471-472)
    std::vector<float> corner_values;    // f at the 8 corners, size 8
    std::vector<float> intersect_coord;  // intersection point per intersected edge, edge order, size 3*n
    std::vector<int> tri_vlist;          // triangles as indices into intersect_coord's points, size 3*num_triangles
    std::vector<int> edge_list;          // the edges that carry an intersection
    float surf_constant = 0.0f;
    int cube_code = 0;                   // extra: the 8-bit configuration (the reference only prints it)
};

// marching.h:26-30
struct Poly_Data {
    std::vector<float> vertex_list;       // point xyz coordinates. size = 3*num_points
    std::vector<unsigned int> tri_list;   // triangle vertex indices, size = num_triangles*3
    std::vector<float> normal_list;       // extra: unit normal per vertex, size = 3*num_points (soup: gradient of f,
                                          // DESIGN.md N1; indexed mesh: CalculateNormal of normal.h)
};

// Source/normal.h:3-41 CalculateNormal: per indexed vertex the sum of cross(B-A, C-A) over its triangles (so
// weighted by triangle area), normalised.  Same float operations in the same order as the glm code the
// reference calls (cross: a.y*b.z - b.y*a.z, ...; normalize: v * (1/sqrt(dot(v,v))), dot = (x*x + y*y) + z*z);
// a vertex used by no triangle, or with a zero sum, comes out NaN exactly as it does there.
inline std::vector<float> CalculateNormal(const Poly_Data* pData) {
    std::vector<float> n(pData->vertex_list.size(), 0.0f);
    const size_t nt = pData->tri_list.size() / 3;
    const float* v = pData->vertex_list.data();
    for (size_t i = 0; i < nt; ++i) {
        const unsigned i1 = pData->tri_list[3 * i], i2 = pData->tri_list[3 * i + 1], i3 = pData->tri_list[3 * i + 2];
        volatile float bax = v[3 * i2] - v[3 * i1], bay = v[3 * i2 + 1] - v[3 * i1 + 1], baz = v[3 * i2 + 2] - v[3 * i1 + 2];
        volatile float cax = v[3 * i3] - v[3 * i1], cay = v[3 * i3 + 1] - v[3 * i1 + 1], caz = v[3 * i3 + 2] - v[3 * i1 + 2];
        volatile float p0 = bay * caz, p1 = cay * baz, p2 = baz * cax, p3 = caz * bax, p4 = bax * cay, p5 = cax * bay;
        const float nx = p0 - p1, ny = p2 - p3, nz = p4 - p5;  // glm::cross(x, y) = (x.y*y.z - y.y*x.z, x.z*y.x - y.z*x.x, x.x*y.y - y.x*x.y)
        const unsigned idx[3] = {i1, i2, i3};
        for (unsigned k : idx) {
            volatile float a = nx + n[3 * k], b = ny + n[3 * k + 1], c = nz + n[3 * k + 2];  // normal + vNormal[i]
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
    return n;
}

class Context {  // one GPU context shared by the facade objects that use it
public:
    explicit Context(int device = 0) {
        if (mc_context_create(device, &h_) != MC_OK) throw std::runtime_error(mc_last_error());
    }
    ~Context() { mc_context_destroy(h_); }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    mc_context* get() const { return h_; }

private:
    mc_context* h_ = nullptr;
};

class Evaluator {
public:
    Evaluator() { set_equation("x+y"); }                       // evaluator.cpp:6-8
    explicit Evaluator(const std::string& s) {                 // evaluator.cpp:10-13
        if (!set_equation(s)) throw std::exception();
    }
    // evaluator.cpp:15-17: on a parse error the previous equation stays in force
    bool set_equation(const std::string& s) {
        if (!mc_expr_check(s.c_str())) return false;
        equation_ = s;
        return true;
    }
    const std::string& equation() const { return equation_; }
    // evaluator.cpp:53; needs a context because the value is computed on the GPU
    float evaluate(Context& ctx, float x, float y, float z) const {
        const float p[3] = {x, y, z};
        float out = 0.0f;
        if (mc_eval_points(ctx.get(), equation_.c_str(), p, 1, &out) != MC_OK) throw std::runtime_error(mc_last_error());
        return out;
    }

private:
    std::string equation_;
};

class Marching {
public:
    explicit Marching(Context& ctx) : ctx_(ctx) {}  // defaults: marching.cpp:23-37

    bool set_evaluator(Evaluator* e) {  // marching.cpp:140-147
        if (!e) return false;
        evaluator_ = e;
        return true;
    }
    bool set_grid_step_size(float v) {  // marching.cpp:226-238
        if (mc_cells_per_axis(v) == 0) return false;
        grid_step_size_ = v;
        return true;
    }
    float get_grid_size() const { return grid_step_size_; }
    void set_surface_constant(float c) { surface_constant_ = c; }  // marching.cpp:149
    void set_scaling_x(float s) { scale_[0] = s; }                 // marching.cpp:240-251
    void set_scaling_y(float s) { scale_[1] = s; }
    void set_scaling_z(float s) { scale_[2] = s; }
    // marching.h:105-113, marching.cpp:173-207.  The constraints live in the GPU context; a cell with a
    // corner outside an enabled constraint is skipped by recalculate() (marching.cpp:476).
    bool set_constraint(int i, const std::string& lhs, const std::string& op, float rhs) {
        return mc_set_constraint(ctx_.get(), i, lhs.c_str(), op.c_str(), rhs) == MC_OK;
    }
    bool set_constraint0(const std::string& l, const std::string& o, float r) { return set_constraint(0, l, o, r); }
    bool set_constraint1(const std::string& l, const std::string& o, float r) { return set_constraint(1, l, o, r); }
    bool set_constraint2(const std::string& l, const std::string& o, float r) { return set_constraint(2, l, o, r); }
    bool use_constraint(int i, bool b) { return mc_use_constraint(ctx_.get(), i, b ? 1 : 0) == MC_OK && b; }
    bool use_constraint0(bool b) { return use_constraint(0, b); }
    bool use_constraint1(bool b) { return use_constraint(1, b); }
    bool use_constraint2(bool b) { return use_constraint(2, b); }
    // marching.cpp:115-137: seed mode keeps the surface reached from the seed's cell (mc_hip.h, mc_set_seed)
    void seed_mode(bool b) { mc_seed_mode(ctx_.get(), b ? 1 : 0); }
    bool set_seed(float x, float y, float z) {
        if (mc_set_seed(ctx_.get(), x, y, z) != MC_OK) return false;
        seed_[0] = x; seed_[1] = y; seed_[2] = z;
        return true;
    }
    void get_seed(float* x, float* y, float* z) const { *x = seed_[0]; *y = seed_[1]; *z = seed_[2]; }
    void want_normals(bool b) { normals_ = b; }

    // calculate_step (marching.cpp:456-595) for the cell with lattice indices (ix, iy, iz) of the current grid, as the
    // reference's step-by-step mode shows it: f comes from the GPU (mc_eval_points), the table walk and the
    // interpolation (marching.cpp:437-446) run here.  false: bad index, no evaluator, or a GPU error.
    bool step_at(int ix, int iy, int iz, Step_Data* out) {
        static const uint64_t tri_row[256] = MC_TRI_ROW_INIT;
        static const uint8_t amb_face[256] = MC_AMB_FACE_INIT;
        static const uint16_t face_corner[6] = MC_FACE_CORNER_INIT;
        static const uint8_t edge_corner[12] = MC_EDGE_CORNER_INIT;
        const int n1 = mc_cells_per_axis(grid_step_size_);
        if (!out || !evaluator_ || ix < 0 || iy < 0 || iz < 0 || ix >= n1 || iy >= n1 || iz >= n1) return false;
        std::vector<float> c((size_t)n1 + 1);
        {
            volatile float v = -1.0f;  // marching.cpp:372-377: the loop variable is advanced by float adds
            for (int i = 0; i <= n1; ++i) {
                c[(size_t)i] = v;
                v = v + grid_step_size_;
            }
        }
        const float x0 = c[ix], x1 = c[ix + 1], y0 = c[iy], y1 = c[iy + 1], z0 = c[iz], z1 = c[iz + 1];
        Step_Data s;
        s.step_i = (iz * n1 + iy) * n1 + ix + 1;
        s.surf_constant = surface_constant_;
        s.corner_coords = {x0, y0, z0, x1, y0, z0, x1, y1, z0, x0, y1, z0, x0, y0, z1, x1, y0, z1, x1, y1, z1, x0, y1, z1};  // :471-472
        s.corner_values.resize(8);
        auto eval = [&](const float* pts, size_t n, float* vals) {  // Marching::evaluate, marching.cpp:209-224
            std::vector<float> sp(3 * n);
            for (size_t i = 0; i < n; ++i)
                for (int a = 0; a < 3; ++a) {
                    volatile float t = scale_[a] * pts[3 * i + a];
                    sp[3 * i + a] = t;
                }
            if (mc_eval_points(ctx_.get(), evaluator_->equation().c_str(), sp.data(), n, vals) == MC_OK) return true;
            error_ = mc_last_error();
            return false;
        };
        if (!eval(s.corner_coords.data(), 8, s.corner_values.data())) return false;
        const float iso = surface_constant_;
        int code = 0;
        for (int i = 0; i < 8; ++i)
            if (s.corner_values[i] > iso) code |= 1 << i;  // :497-505
        s.cube_code = code;
        *out = s;
        if (code == 0 || code == 255) return true;  // :508-510
        int row = code;
        if (amb_face[code] != 0xFF) {  // :523-549
            volatile float m[3] = {0.0f, 0.0f, 0.0f};
            for (int i = 0; i < 4; ++i) {
                const int vi = (face_corner[amb_face[code]] >> (4 * i)) & 0xF;
                for (int a = 0; a < 3; ++a) m[a] = m[a] + s.corner_coords[3 * vi + a];
            }
            float mid[3] = {(float)((double)m[0] / 4.0), (float)((double)m[1] / 4.0), (float)((double)m[2] / 4.0)}, mv = 0.0f;
            if (!eval(mid, 1, &mv)) return false;
            if (mv > iso) row = 255 - code;
        }
        auto interp = [&](float xs, float xe, float vs, float ve) {  // :437-446
            volatile float q = (iso - vs) / (ve - vs);
            volatile float v = q * (xe - xs);
            if (std::isinf((float)v) || std::isnan((float)v)) return (float)((double)xs + 0.5 * (double)(xe - xs));
            volatile float r = xs + v;
            return (float)r;
        };
        int mapper[12];
        for (int e = 0; e < 12; ++e) {  // :557-583
            mapper[e] = 12;
            const int v1 = edge_corner[e] & 0xF, v2 = edge_corner[e] >> 4;
            if (((code >> v1) & 1) != ((code >> v2) & 1)) {
                out->edge_list.push_back(e);
                mapper[e] = (int)(out->intersect_coord.size() / 3);
                for (int a = 0; a < 3; ++a)
                    out->intersect_coord.push_back(interp(s.corner_coords[3 * v1 + a], s.corner_coords[3 * v2 + a], s.corner_values[v1],
                                                          s.corner_values[v2]));
            }
        }
        for (int k = 0; k < 15; k += 3) {  // :586-594
            const int e1 = (int)((tri_row[row] >> (4 * k)) & 0xF), e2 = (int)((tri_row[row] >> (4 * k + 4)) & 0xF),
                      e3 = (int)((tri_row[row] >> (4 * k + 8)) & 0xF);
            if (e1 == 0xF) break;
            out->tri_vlist.push_back(mapper[e1]);
            out->tri_vlist.push_back(mapper[e2]);
            out->tri_vlist.push_back(mapper[e3]);
        }
        return true;
    }
    // true: weld vertices like the reference (marching.cpp:599-654); false (default): triangle soup
    void set_indexed(bool b) { indexed_ = b; }

    void reset_all_data() {  // marching.cpp:293-305
        poly_data_.vertex_list.clear();
        poly_data_.tri_list.clear();
        poly_data_.normal_list.clear();
    }

    // marching.cpp:308, full-sweep branch :368-384.  false = no evaluator or a GPU/compile error.
    bool recalculate() {
        reset_all_data();
        error_.clear();
        if (!evaluator_) return false;
        mc_params p{};
        p.equation = evaluator_->equation().c_str();
        p.step = grid_step_size_;
        p.iso = surface_constant_;
        p.scale[0] = scale_[0];
        p.scale[1] = scale_[1];
        p.scale[2] = scale_[2];
        p.flags = (normals_ && !indexed_ ? MC_FLAG_NORMALS : 0u) | (indexed_ ? MC_FLAG_TRI_META : 0u);
        p.z_begin = 0;
        p.z_end = -1;
        mc_result r{};
        if (mc_march(ctx_.get(), &p, &r) != MC_OK) {
            error_ = mc_last_error();
            return false;
        }
        last_ = r;
        const size_t nv = (size_t)r.n_tris * 3;
        std::vector<float> inter(nv * 6);
        if (nv && mc_copy_vertices(ctx_.get(), inter.data(), r.n_tris) != MC_OK) {
            error_ = mc_last_error();
            return false;
        }
        if (indexed_) return weld_like_reference(inter, r.n_tris);
        poly_data_.vertex_list.resize(nv * 3);
        poly_data_.normal_list.resize(nv * 3);
        poly_data_.tri_list.resize(nv);
        for (size_t i = 0; i < nv; ++i) {
            for (int k = 0; k < 3; ++k) {
                poly_data_.vertex_list[3 * i + k] = inter[6 * i + k];
                poly_data_.normal_list[3 * i + k] = inter[6 * i + 3 + k];
            }
            poly_data_.tri_list[i] = (unsigned int)i;
        }
        return true;
    }

    const Poly_Data* get_poly_data() const { return &poly_data_; }  // marching.cpp:656-658

    // marching.cpp:771-854 save_poly_to_file: the same ASCII PLY, byte for byte ("element face N "
    // carries the reference's trailing blank), to an explicit path instead of a Win32 dialog
    bool save_poly_to_file(const std::string& path) const {
        if (poly_data_.vertex_list.empty()) return false;
        FILE* fp = std::fopen(path.c_str(), "w");
        if (!fp) return false;
        const int np = (int)(poly_data_.vertex_list.size() / 3), nt = (int)(poly_data_.tri_list.size() / 3);
        std::fprintf(fp, "ply\nformat ascii 1.0\n");
        std::fprintf(fp, "element vertex %d\n", np);
        std::fprintf(fp, "property float x\nproperty float y\nproperty float z\n");
        std::fprintf(fp, "element face %d \n", nt);
        std::fprintf(fp, "property list uchar int vertex_indices\nend_header\n");
        for (int i = 0; i < np; ++i)
            std::fprintf(fp, "%f %f %f\n", poly_data_.vertex_list[3 * i], poly_data_.vertex_list[3 * i + 1],
                         poly_data_.vertex_list[3 * i + 2]);
        for (int i = 0; i < nt; ++i)
            std::fprintf(fp, "%u %u %u %u\n", 3u, poly_data_.tri_list[3 * i], poly_data_.tri_list[3 * i + 1],
                         poly_data_.tri_list[3 * i + 2]);
        std::fclose(fp);
        return true;
    }

    // marching.cpp:665-768 load_poly_from_file: same header rules ("ply", "format ascii 1.0", at most
    // 10 header lines, "element vertex N", "element face N") and the same append-to-current-data
    // behaviour (the reference does not clear poly_data first)
    bool load_poly_from_file(const std::string& path) {
        FILE* fp = std::fopen(path.c_str(), "r");
        if (!fp) return false;
        char line[256];
        auto getl = [&](std::string& out) {
            if (!std::fgets(line, sizeof line, fp)) { out.clear(); return false; }
            out = line;
            return true;
        };
        std::string str;
        if (!getl(str) || str.find("ply") == std::string::npos) { std::fclose(fp); return false; }
        if (!getl(str) || str.find("format ascii 1.0") == std::string::npos) { std::fclose(fp); return false; }
        getl(str);
        int counter = 2, np = 0, nt = 0;
        while (str.find("end_header") == std::string::npos) {
            if (++counter >= 10) { std::fclose(fp); return false; }
            if (str.find("element vertex") != std::string::npos) np = std::atoi(str.c_str() + 15);
            if (str.find("element face") != std::string::npos) nt = std::atoi(str.c_str() + 13);
            if (str.empty()) break;
            getl(str);
        }
        for (int i = 0; i < np; ++i) {
            float x, y, z;
            if (std::fscanf(fp, "%f", &x) == EOF || std::fscanf(fp, "%f", &y) == EOF || std::fscanf(fp, "%f", &z) == EOF) {
                std::fclose(fp);
                return false;
            }
            poly_data_.vertex_list.push_back(x);
            poly_data_.vertex_list.push_back(y);
            poly_data_.vertex_list.push_back(z);
        }
        for (int i = 0; i < nt; ++i) {
            unsigned num = 0, a, b, c;
            if (std::fscanf(fp, "%u %u %u %u", &num, &a, &b, &c) == EOF) return false;
            if (num != 3) { std::fclose(fp); return false; }
            poly_data_.tri_list.push_back(a);
            poly_data_.tri_list.push_back(b);
            poly_data_.tri_list.push_back(c);
        }
        std::fclose(fp);
        return true;
    }
    const mc_result& last_result() const { return last_; }
    const std::string& last_error() const { return error_; }

private:
    // marching.h:32-55: the reference's point type; its operator< treats coordinates closer than 1e-6
    // as equal, axis by axis (not a strict weak order -- reproduced as is, with the same container)
    struct xyz {
        float x, y, z;
        int idx;
        static bool close_enough(float a, float b) { return std::fabs(a - b) < 0.000001; }
        bool operator<(const xyz& r) const {
            if (!close_enough(x, r.x)) return x < r.x;
            if (!close_enough(y, r.y)) return y < r.y;
            if (!close_enough(z, r.z)) return z < r.z;
            return false;
        }
    };

    // marching.cpp:599-654 on the soup: per cell, the crossed-edge vertices are inserted in edge
    // order 0..11 (the order calculate_step builds intersect_coord, :557-583), then the cell's
    // triangles are appended with the returned indices.
    bool weld_like_reference(const std::vector<float>& inter, uint64_t n_tris) {
        static const uint64_t tri_row[256] = MC_TRI_ROW_INIT;
        std::vector<uint16_t> meta(n_tris);
        if (n_tris && mc_copy_tri_meta(ctx_.get(), meta.data(), n_tris) != MC_OK) {
            error_ = mc_last_error();
            return false;
        }
        std::set<xyz> vertex_set;
        uint64_t t0 = 0;
        while (t0 < n_tris) {
            uint64_t t1 = t0 + 1;
            while (t1 < n_tris && (meta[t1] >> 8) != 0) ++t1;  // triangles t0..t1-1 belong to one cell
            const int row = meta[t0] & 0xFF;
            const float* pos[12] = {nullptr};
            for (uint64_t t = t0; t < t1; ++t)
                for (int k = 0; k < 3; ++k) {
                    const int e = (int)((tri_row[row] >> (4 * (3 * (int)(t - t0) + k))) & 0xF);
                    pos[e] = &inter[(t * 3 + k) * 6];
                }
            int vidx[12];
            for (int e = 0; e < 12; ++e) {
                vidx[e] = -1;
                if (pos[e] && !std::isnan(pos[e][0])) {  // :611-613 NaN x is skipped, index stays -1
                    const int new_i = (int)(poly_data_.vertex_list.size() / 3);  // :629
                    const int found = vertex_set.insert(xyz{pos[e][0], pos[e][1], pos[e][2], new_i}).first->idx;
                    if (found == new_i) {
                        poly_data_.vertex_list.push_back(pos[e][0]);
                        poly_data_.vertex_list.push_back(pos[e][1]);
                        poly_data_.vertex_list.push_back(pos[e][2]);
                    }
                    vidx[e] = found;
                }
            }
            for (uint64_t t = t0; t < t1; ++t)
                for (int k = 0; k < 3; ++k) {
                    const int e = (int)((tri_row[row] >> (4 * (3 * (int)(t - t0) + k))) & 0xF);
                    poly_data_.tri_list.push_back((unsigned int)vidx[e]);
                }
            t0 = t1;
        }
        if (normals_) poly_data_.normal_list = CalculateNormal(&poly_data_);  // what the reference's drawer computes
        return true;
    }

    Context& ctx_;
    bool indexed_ = false;
    Evaluator* evaluator_ = nullptr;   // borrowed, never owned (marching.cpp:140-147)
    float grid_step_size_ = 0.25f;     // marching.cpp:24
    float surface_constant_ = 0.0f;
    float scale_[3] = {1.0f, 1.0f, 1.0f};
    bool normals_ = true;
    float seed_[3] = {0.0f, 0.0f, 0.0f};  // marching.cpp:35
    Poly_Data poly_data_;
    mc_result last_{};
    std::string error_;
};

}  // namespace mc_amd
