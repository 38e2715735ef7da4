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
// Differences, all documented in DESIGN.md: the mesh is triangle SOUP (tri_list = 0..3T-1, the
// reference's std::set vertex welding, marching.cpp:627-643, is a "next" row); `normal_list`
// (gradient normals, 3 floats per vertex) is an extra member; step-by-step, seed mode,
// constraints and PLY dialogs are outside the hot path and not provided; a failed GPU call makes
// recalculate() return false and last_error() non-empty instead of crashing.
#pragma once
#include <cstdint>
#include <exception>
#include <stdexcept>
#include <string>
#include <vector>

#include "mc_hip.h"

namespace mc_amd {

// marching.h:26-30 (Step_Data, the one-cell teaching trace, is not part of the hot path)
struct Poly_Data {
    std::vector<float> vertex_list;       // point xyz coordinates. size = 3*num_points
    std::vector<unsigned int> tri_list;   // triangle vertex indices, size = num_triangles*3
    std::vector<float> normal_list;       // extra: unit gradient normal per vertex. size = 3*num_points
};

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
    void want_normals(bool b) { normals_ = b; }

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
        p.flags = normals_ ? MC_FLAG_NORMALS : 0u;
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
    const mc_result& last_result() const { return last_; }
    const std::string& last_error() const { return error_; }

private:
    Context& ctx_;
    Evaluator* evaluator_ = nullptr;   // borrowed, never owned (marching.cpp:140-147)
    float grid_step_size_ = 0.25f;     // marching.cpp:24
    float surface_constant_ = 0.0f;
    float scale_[3] = {1.0f, 1.0f, 1.0f};
    bool normals_ = true;
    Poly_Data poly_data_;
    mc_result last_{};
    std::string error_;
};

}  // namespace mc_amd
