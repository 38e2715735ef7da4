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
// Differences, all documented in DESIGN.md.  get_poly_data() holds the reference's INDEXED mesh (vertex_list welded,
// tri_list indexing it), built on the GPU with a closed form of the reference's welding rule (marching.cpp:599-654,
// marching.h:32-55: first point inserted wins, 1e-6 tolerance; MC_FLAG_INDEXED -- identical to the reference's std::set
// except where points lie within 1e-6 of each other without being bit-identical, which are always merged here: mc_hip.h),
// plus `normal_list`, an extra member: the drawer's
// area-weighted vertex normals (CalculateNormal, Source/normal.h:3-41, also available as a free function).
// set_indexed(false) hands over the GPU's triangle SOUP instead (tri_list = 0..3T-1) with gradient normals.  Constraints
// (set_constraint0..2 / use_constraint0..2) and seed mode (seed_mode / set_seed) are provided -- seed mode returns the same
// triangles as the reference's walk, welded like the dense sweep's, in sweep order instead of visitation order (mc_hip.h).
// The step-by-step STATE MACHINE of
// recalculate() is not; step_at(ix, iy, iz) gives the Step_Data of any one cell instead.  A failed GPU call makes
// recalculate() return false and last_error() non-empty instead of crashing.
//
// Evaluator() / Marching() work like the reference's (evaluator.h:61, marching.h:75: no arguments): they share one
// process-wide GPU context on device 0, created on first use.  The Context& overloads put an object on another GPU,
// and Marching::set_devices({0, 1, ...}) spreads ONE recalculate() over several: the cell layers are cut into one Z slab
// per listed device, swept at once (mc_march_sharded), and get_poly_data() holds the same Poly_Data as a single sweep.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <exception>
#include <cstdlib>
#include <cstring>
#include <memory>
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
    Step_Data step_data;                  // marching.h:29: one cell's evaluation; filled by Marching::step_at (the sweep
                                          // itself never builds it: the reference fills it per cell, marching.cpp:456-595)
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

// the process-wide context the argument-less constructors use (device 0, created on first use)
inline Context& default_context() {
    static Context ctx(0);
    return ctx;
}

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
    // evaluator.cpp:53 / evaluator.h:61; the value is computed on the GPU (the process-wide context, or the one given)
    float evaluate(float x, float y, float z) const { return evaluate(default_context(), x, y, z); }
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
    Marching() : ctx_(default_context()) {}         // marching.h:75; defaults: marching.cpp:23-37
    explicit Marching(Context& ctx) : ctx_(ctx) {}  // the same on a chosen GPU

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
    // marching.h:105-113, marching.cpp:173-207.  Constraints, seed mode and the seed belong to THIS object, as in the
    // reference (marching.h:58-69, :130-157); the GPU context several objects may share only holds the compiled copy of
    // whichever object swept last -- recalculate() and step_at() push this object's state first (push_state).  A cell
    // with a corner outside an enabled constraint is skipped by recalculate() (marching.cpp:476).
    bool set_constraint(int i, const std::string& lhs, const std::string& op, float rhs) {
        if (i < 0 || i > 2) return false;
        if (mc_set_constraint(ctx_.get(), i, lhs.c_str(), op.c_str(), rhs) != MC_OK) return false;
        cons_[i].valid = true;
        cons_[i].lhs = lhs;
        cons_[i].ops = op;
        cons_[i].op = op == ">=" ? 0 : op == "<=" ? 1 : op == ">" ? 2 : 3;
        cons_[i].rhs = rhs;
        return true;
    }
    bool set_constraint0(const std::string& l, const std::string& o, float r) { return set_constraint(0, l, o, r); }
    bool set_constraint1(const std::string& l, const std::string& o, float r) { return set_constraint(1, l, o, r); }
    bool set_constraint2(const std::string& l, const std::string& o, float r) { return set_constraint(2, l, o, r); }
    bool use_constraint(int i, bool b) {
        if (i < 0 || i > 2) return false;
        cons_[i].in_use = b;
        return b;
    }
    bool use_constraint0(bool b) { return use_constraint(0, b); }
    bool use_constraint1(bool b) { return use_constraint(1, b); }
    bool use_constraint2(bool b) { return use_constraint(2, b); }
    // marching.cpp:115-137: seed mode keeps the surface reached from the seed's cell (mc_hip.h, mc_set_seed)
    void seed_mode(bool b) { seed_mode_ = b; }
    bool set_seed(float x, float y, float z) {
        if (!((x <= 1 && x >= -1) && (y >= -1 && y <= 1) && (z >= -1 && z <= 1))) return false;  // marching.cpp:128
        seed_[0] = x; seed_[1] = y; seed_[2] = z;
        return true;
    }
    void get_seed(float* x, float* y, float* z) const { *x = seed_[0]; *y = seed_[1]; *z = seed_[2]; }
    void want_normals(bool b) { normals_ = b; }

    // calculate_step (marching.cpp:456-595) for the cell with lattice indices (ix, iy, iz) of the current grid, as the
    // reference's step-by-step mode shows it: f comes from the GPU (mc_eval_points), the table walk and the
    // interpolation (marching.cpp:437-446) run here.  false: bad index, no evaluator, or a GPU error.
    bool step_at(int ix, int iy, int iz, Step_Data* out) {
        const int n1 = mc_cells_per_axis(grid_step_size_);
        if (!out || !evaluator_ || ix < 0 || iy < 0 || iz < 0 || ix >= n1 || iy >= n1 || iz >= n1) return false;
        if (!step_at_impl(ix, iy, iz, n1, out)) return false;
        poly_data_.step_data = *out;  // marching.h:29: where the reference's callers read the cell's trace
        return true;
    }

private:
    bool step_at_impl(int ix, int iy, int iz, int n1, Step_Data* out) {
        static const uint64_t tri_row[256] = MC_TRI_ROW_INIT;
        static const uint8_t amb_face[256] = MC_AMB_FACE_INIT;
        static const uint16_t face_corner[6] = MC_FACE_CORNER_INIT;
        static const uint8_t edge_corner[12] = MC_EDGE_CORNER_INIT;
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
        // marching.cpp:476: a corner outside an enabled constraint abandons the cell before anything is computed
        for (int ci = 0; ci < 3; ++ci) {
            const Constraint& cn = cons_[ci];
            if (!(cn.valid && cn.in_use)) continue;  // marching.cpp:258
            float lhs[8];
            std::vector<float> sp(24);
            for (int i = 0; i < 8; ++i)
                for (int a = 0; a < 3; ++a) {
                    volatile float t = scale_[a] * s.corner_coords[3 * i + a];
                    sp[3 * i + a] = t;
                }
            if (mc_eval_points(ctx_.get(), cn.lhs.c_str(), sp.data(), 8, lhs) != MC_OK) {
                error_ = mc_last_error();
                return false;
            }
            for (int i = 0; i < 8; ++i) {
                const bool ok = cn.op == 0 ? lhs[i] >= cn.rhs : cn.op == 1 ? lhs[i] <= cn.rhs : cn.op == 2 ? lhs[i] > cn.rhs : lhs[i] < cn.rhs;
                if (!ok) {
                    *out = s;  // an empty Step_Data: coordinates only
                    return true;
                }
            }
        }
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

    // This object's constraints, seed mode and seed -> the (possibly shared) GPU context.  The library ignores calls
    // that change nothing, so this is cheap when one object sweeps repeatedly.
    bool push_state(mc_context* c) {
        for (int i = 0; i < 3; ++i) {
            const Constraint& cn = cons_[i];
            if (cn.valid && mc_set_constraint(c, i, cn.lhs.c_str(), cn.ops.c_str(), cn.rhs) != MC_OK) return false;
            if (mc_use_constraint(c, i, (cn.valid && cn.in_use) ? 1 : 0) != MC_OK) return false;  // marching.cpp:258
        }
        if (mc_set_seed(c, seed_[0], seed_[1], seed_[2]) != MC_OK) return false;
        return mc_seed_mode(c, seed_mode_ ? 1 : 0) == MC_OK;
    }
    bool push_state() {
        if (!push_state(ctx_.get())) return false;
        for (auto& c : shard_ctx_)
            if (!push_state(c->get())) return false;
        return true;
    }

public:
    // The device list of this object's sweeps: recalculate() cuts the grid's cell layers into one contiguous
    // Z slab per entry and sweeps them at once, one context per entry (a device may be listed more than once).  The mesh
    // get_poly_data() returns is the single sweep's -- the sweep is z-major (marching.cpp:375), so the slabs' lists
    // concatenate to it, and the indexed mesh is welded across the seams (mc_hip.h: mc_march_sharded).  An empty list
    // returns to the one context the object was constructed on.  Seed mode needs the whole grid on one device: while it is
    // on, the first device of the list sweeps everything (the mesh is the same).  false: a device of the list cannot be used.
    bool set_devices(const std::vector<int>& devices) {
        std::vector<std::unique_ptr<Context>> made;
        try {
            for (int d : devices) made.emplace_back(new Context(d));
        } catch (const std::exception& e) {
            error_ = e.what();
            return false;
        }
        shard_ctx_ = std::move(made);
        return true;
    }
    size_t device_count() const { return shard_ctx_.empty() ? 1 : shard_ctx_.size(); }
    // the slabs of the last multi-device recalculate(): layers and offsets per device of the list
    const std::vector<mc_shard>& shards() const { return shards_; }

    // true (default): the reference's welded Poly_Data (marching.cpp:599-654), built on the GPU; false: triangle soup
    void set_indexed(bool b) { indexed_ = b; }

    void reset_all_data() {  // marching.cpp:293-305
        iota_upto_ = 0;
        poly_data_.vertex_list.clear();
        poly_data_.tri_list.clear();
        poly_data_.normal_list.clear();
        poly_data_.step_data.intersect_coord.clear();  // marching.cpp:296-298
        poly_data_.step_data.tri_vlist.clear();
        poly_data_.step_data.edge_list.clear();
        poly_data_.step_data.step_i = -2;              // reset_step, marching.cpp:288-290
    }

    // marching.cpp:308, full-sweep branch :368-384.  false = no evaluator or a GPU/compile error.
    bool recalculate() {
        // marching.cpp:293-305 reset_all_data: the mesh is gone until the sweep has delivered.  The big vectors are NOT
        // cleared up front -- clear() + resize() would value-initialise (memset) every element again, 180 MB per sweep at
        // grid_res 1024, tens of milliseconds of host time around a 1.6 ms GPU job; they are resized in place (free when the
        // counts repeat, as in an animation) and emptied only if the sweep fails.
        struct OnFailure {
            Poly_Data& pd;
            bool ok = false;
            ~OnFailure() {
                if (!ok) {
                    pd.vertex_list.clear();
                    pd.tri_list.clear();
                    pd.normal_list.clear();
                }
            }
        } guard{poly_data_};
        poly_data_.step_data.intersect_coord.clear();  // marching.cpp:296-298
        poly_data_.step_data.tri_vlist.clear();
        poly_data_.step_data.edge_list.clear();
        poly_data_.step_data.step_i = -2;              // reset_step, marching.cpp:288-290
        error_.clear();
        const bool done = recalculate_impl();
        guard.ok = done;
        return done;
    }

private:
    bool recalculate_impl() {
        if (!evaluator_) return false;
        if (!push_state()) {
            error_ = mc_last_error();
            return false;
        }
        const bool indexed = indexed_;  // (seed mode too: marching.cpp:310-331 feeds add_step_to_poly_data like the dense sweep)
        mc_params p{};
        p.equation = evaluator_->equation().c_str();
        p.step = grid_step_size_;
        p.iso = surface_constant_;
        p.scale[0] = scale_[0];
        p.scale[1] = scale_[1];
        p.scale[2] = scale_[2];
        p.flags = indexed ? (MC_FLAG_INDEXED | MC_FLAG_NO_EMIT) : (normals_ ? MC_FLAG_NORMALS : 0u);
        p.z_begin = 0;
        p.z_end = -1;
        if (!shard_ctx_.empty()) return recalculate_sharded(p, indexed);
        mc_result r{};
        if (mc_march(ctx_.get(), &p, &r) != MC_OK) {
            error_ = mc_last_error();
            return false;
        }
        last_ = r;
        if (indexed) {
            poly_data_.vertex_list.resize((size_t)r.n_verts * 3);
            poly_data_.tri_list.resize((size_t)r.n_tris * 3);
            poly_data_.normal_list.resize(normals_ ? (size_t)r.n_verts * 3 : 0);
            iota_upto_ = 0;
            static_assert(sizeof(unsigned int) == sizeof(uint32_t), "tri_list is handed to the GPU library as uint32");
            if (mc_copy_indexed(ctx_.get(), poly_data_.vertex_list.data(), reinterpret_cast<uint32_t*>(poly_data_.tri_list.data()),
                                normals_ ? poly_data_.normal_list.data() : nullptr, r.n_verts, r.n_tris) != MC_OK) {
                error_ = mc_last_error();
                return false;
            }
            return true;
        }
        // soup: positions and normals are split on the GPU and land in Poly_Data's vectors directly (no staging vector, no
        // host loop over 30 M vertices); tri_list = 0 .. 3T-1
        const size_t nv = (size_t)r.n_tris * 3;
        poly_data_.vertex_list.resize(nv * 3);
        poly_data_.normal_list.resize(nv * 3);
        if (nv && (mc_copy_soup(ctx_.get(), poly_data_.vertex_list.data(), r.n_tris) != MC_OK ||
                   mc_copy_soup_normals(ctx_.get(), poly_data_.normal_list.data(), r.n_tris) != MC_OK)) {
            error_ = mc_last_error();
            return false;
        }
        fill_iota(nv);
        return true;
    }
    // tri_list = 0, 1, 2, ... (soup); written only where it is not that already
    void fill_iota(size_t nv) {
        std::vector<unsigned int>& t = poly_data_.tri_list;
        const bool was_iota = iota_upto_ > 0 && t.size() >= 1 && t.size() <= iota_upto_;
        const size_t keep = was_iota ? (t.size() < nv ? t.size() : nv) : 0;
        t.resize(nv);
        for (size_t i = keep; i < nv; ++i) t[i] = (unsigned int)i;
        iota_upto_ = nv;
    }

    // recalculate() over the device list: one slab per context, all at once; the hand-over is the single sweep's
    bool recalculate_sharded(const mc_params& p, bool indexed) {
        const int n = (int)shard_ctx_.size();
        std::vector<mc_context*> cs;
        for (auto& c : shard_ctx_) cs.push_back(c->get());
        std::vector<mc_result> rs((size_t)n);
        shards_.assign((size_t)n, mc_shard{});
        if (mc_march_sharded(cs.data(), n, &p, nullptr, rs.data(), shards_.data()) != MC_OK) {
            error_ = mc_last_error();
            return false;
        }
        last_ = rs[0];
        uint64_t nt = 0, nv = 0, nc = 0, na = 0;
        for (const mc_result& r : rs) {
            nt += r.n_tris;
            nv += r.n_verts;
            nc += r.n_cells;
            na += r.n_active;
        }
        last_.z_end = rs[(size_t)n - 1].z_end;
        last_.n_tris = nt;
        last_.n_verts = nv;
        last_.n_cells = nc;
        last_.n_active = na;
        last_.d_vertices = nullptr;  // (the device buffers are per slab: results of one context only describe its slab)
        last_.d_codes = nullptr;
        last_.d_codes_tail = nullptr;
        last_.d_vertex_list = nullptr;
        last_.d_tri_list = nullptr;
        last_.d_vertex_normals = nullptr;
        last_.d_totals = nullptr;
        if (indexed) {
            poly_data_.vertex_list.resize((size_t)nv * 3);
            poly_data_.tri_list.resize((size_t)nt * 3);
            poly_data_.normal_list.resize(normals_ ? (size_t)nv * 3 : 0);
            iota_upto_ = 0;
            if (mc_copy_sharded_indexed(cs.data(), n, poly_data_.vertex_list.data(), reinterpret_cast<uint32_t*>(poly_data_.tri_list.data()),
                                        normals_ ? poly_data_.normal_list.data() : nullptr, nv, nt) != MC_OK) {
                error_ = mc_last_error();
                return false;
            }
            return true;
        }
        const size_t nvert = (size_t)nt * 3;
        poly_data_.vertex_list.resize(nvert * 3);
        poly_data_.normal_list.resize(nvert * 3);
        for (int i = 0; i < n; ++i) {  // every slab's halves straight to their place in the whole list
            const size_t at = (size_t)shards_[(size_t)i].tri_offset * 9;
            if (rs[(size_t)i].n_tris && (mc_copy_soup(cs[(size_t)i], poly_data_.vertex_list.data() + at, rs[(size_t)i].n_tris) != MC_OK ||
                                         mc_copy_soup_normals(cs[(size_t)i], poly_data_.normal_list.data() + at, rs[(size_t)i].n_tris) != MC_OK)) {
                error_ = mc_last_error();
                return false;
            }
        }
        fill_iota(nvert);
        return true;
    }

public:
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

    // Reader of the format save_poly_to_file writes (marching.cpp:817-851), with the reference loader's rules
    // (marching.cpp:695-768): first line "ply", second "format ascii 1.0", the counts from the "element vertex N" /
    // "element face N" lines of a header of at most 10 lines up to "end_header", then N x "x y z" and N x "3 i j k";
    // the mesh is APPENDED to the current one (the reference does not clear poly_data either).  Any violation -> false.
    bool load_poly_from_file(const std::string& path) {
        struct File {
            FILE* f;
            explicit File(const char* p) : f(std::fopen(p, "r")) {}
            ~File() { if (f) std::fclose(f); }
        } in(path.c_str());
        if (!in.f) return false;
        char buf[256];
        auto next_line = [&](std::string& line) {
            line.clear();
            if (std::fgets(buf, sizeof buf, in.f)) line = buf;
        };
        auto count_after = [](const std::string& line, const char* tag, long& n) {
            const size_t at = line.find(tag);
            if (at != std::string::npos) n = std::strtol(line.c_str() + at + std::strlen(tag), nullptr, 10);
        };
        std::string line;
        next_line(line);
        if (line.find("ply") == std::string::npos) return false;
        next_line(line);
        if (line.find("format ascii 1.0") == std::string::npos) return false;
        long n_points = 0, n_faces = 0;
        bool ended = false;
        for (int header_line = 3; header_line <= 10 && !ended; ++header_line) {  // lines 3..10 of the file
            next_line(line);
            if (line.empty()) break;  // end of file inside the header: no data follows, counts as read so far
            if (line.find("end_header") != std::string::npos) ended = true;
            else {
                count_after(line, "element vertex", n_points);
                count_after(line, "element face", n_faces);
            }
        }
        if (!ended && !line.empty()) return false;  // a header longer than the reference accepts
        std::vector<float> pts;
        std::vector<unsigned int> tri;
        for (long i = 0; i < n_points; ++i) {
            float v[3];
            if (std::fscanf(in.f, "%f %f %f", &v[0], &v[1], &v[2]) != 3) return false;
            pts.insert(pts.end(), v, v + 3);
        }
        for (long i = 0; i < n_faces; ++i) {
            unsigned int k = 0, a = 0, b = 0, c = 0;
            if (std::fscanf(in.f, "%u %u %u %u", &k, &a, &b, &c) != 4 || k != 3) return false;
            tri.push_back(a);
            tri.push_back(b);
            tri.push_back(c);
        }
        poly_data_.vertex_list.insert(poly_data_.vertex_list.end(), pts.begin(), pts.end());
        poly_data_.tri_list.insert(poly_data_.tri_list.end(), tri.begin(), tri.end());
        iota_upto_ = 0;
        return true;
    }
    const mc_result& last_result() const { return last_; }
    const std::string& last_error() const { return error_; }

private:
    struct Constraint {  // marching.h:58-69 (the GPU context holds the compiled copy)
        bool valid = false, in_use = false;
        std::string lhs, ops;
        int op = 0;  // 0 '>=', 1 '<=', 2 '>', 3 '<' (ops: as spelled)
        float rhs = 0.0f;
    } cons_[3];

    size_t iota_upto_ = 0;  // tri_list[0 .. iota_upto_) is known to be 0, 1, 2, ... (the soup hand-over wrote it; 0: unknown)
    Context& ctx_;
    std::vector<std::unique_ptr<Context>> shard_ctx_;  // set_devices: one context per listed device (empty: ctx_ alone)
    std::vector<mc_shard> shards_;
    bool indexed_ = true;
    bool seed_mode_ = false;
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
