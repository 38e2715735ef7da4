// sharded_caller.cpp -- a reference-style caller (Source/main.cpp:11-14, Source/marching_test_drawer.h:7-15) that gives its
// Marching object a DEVICE LIST and checks, in C++ and through the C ABI alone, that the mesh is the one a
// single sweep hands out, bit for bit: welded Poly_Data (vertex_list, tri_list, CalculateNormal) and triangle soup, with
// and without a constraint, for 2 / 3 / 4 / 8 entries.  On a one-GPU box the list names device 0 several times
// (argv[1] = comma-separated device list overrides that, e.g. "0,1,2,3" on a multi-GPU node).
// Prints SHARDED_OK and one line per case; exit code 0 only if every comparison held.
#include "marching.h"
#include "evaluator.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <execinfo.h>
#include <signal.h>
#include <unistd.h>

// a crash in here must say where (the test shows this program's output): frames to stderr, then the default action
static void on_crash(int sig) {
    void* frames[64];
    const int n = backtrace(frames, 64);
    const char msg[] = "sharded_caller: fatal signal, backtrace:\n";
    (void)!write(2, msg, sizeof msg - 1);
    backtrace_symbols_fd(frames, n, 2);
    signal(sig, SIG_DFL);
    raise(sig);
}

static bool same_bits(const std::vector<float>& a, const std::vector<float>& b) {
    return a.size() == b.size() && (a.empty() || std::memcmp(a.data(), b.data(), a.size() * sizeof(float)) == 0);
}

struct Mesh {
    std::vector<float> v, n;
    std::vector<unsigned int> t;
};

static bool sweep(Marching& m, bool indexed, Mesh& out) {
    m.set_indexed(indexed);
    if (!m.recalculate()) {
        std::printf("error: %s\n", m.last_error().c_str());
        return false;
    }
    const Poly_Data* p = m.get_poly_data();
    out.v = p->vertex_list;
    out.t = p->tri_list;
    out.n = p->normal_list;
    return true;
}

int main(int argc, char** argv) {
    signal(SIGSEGV, on_crash);
    signal(SIGABRT, on_crash);
    signal(SIGBUS, on_crash);
    setvbuf(stdout, nullptr, _IOLBF, 0);
    std::vector<int> base;
    if (argc > 1)
        for (const char* q = argv[1]; q && *q;) {
            base.push_back(std::atoi(q));
            q = std::strchr(q, ',');
            if (q) ++q;
        }
    struct Case {
        const char* eq;
        float step, iso, sx;
        const char* cons;  // "" or a constraint lhs for `lhs > -0.25`
    } cases[] = {
        {"x^2+y^2+z^2-1", 2.0f / 48, 0.0f, 1.0f, ""},
        {"(x^2)^2+(y^2)^2+(z^2)^2-(x^2+y^2+z^2)", 2.0f / 40, -0.4f, 1.0f, ""},
        {"x^2+y^2+z^2-1", 0.07f, 0.0f, 1.1f, "z+0.3*x"},
        {"x+y", 2.0f / 32, 0.0f, 1.0f, ""},
        {"z", 2.0f / 16, 0.0f, 1.0f, ""},  // the surface lies IN lattice planes: every seam vertex is a lattice hit
    };
    int bad = 0;
    for (const Case& c : cases) {
        Evaluator ev;
        if (!ev.set_equation(c.eq)) return 2;
        Marching whole;
        whole.set_evaluator(&ev);
        whole.set_grid_step_size(c.step);
        whole.set_surface_constant(c.iso);
        whole.set_scaling_x(c.sx);
        if (*c.cons) {
            if (!whole.set_constraint0(c.cons, ">", -0.25f)) return 3;
            whole.use_constraint0(true);
        }
        Mesh wi, ws;
        if (!sweep(whole, true, wi) || !sweep(whole, false, ws)) return 4;
        for (int n : {2, 3, 4, 8}) {
            std::vector<int> devs;
            for (int i = 0; i < n; ++i) devs.push_back(base.empty() ? 0 : base[(size_t)i % base.size()]);
            Marching parts;
            parts.set_evaluator(&ev);
            parts.set_grid_step_size(c.step);
            parts.set_surface_constant(c.iso);
            parts.set_scaling_x(c.sx);
            if (*c.cons) {
                parts.set_constraint0(c.cons, ">", -0.25f);
                parts.use_constraint0(true);
            }
            if (!parts.set_devices(devs)) {
                std::printf("set_devices failed: %s\n", parts.last_error().c_str());
                return 5;
            }
            Mesh pi, ps;
            if (!sweep(parts, true, pi) || !sweep(parts, false, ps)) return 6;
            const bool ok_i = same_bits(wi.v, pi.v) && wi.t == pi.t && same_bits(wi.n, pi.n);
            const bool ok_s = same_bits(ws.v, ps.v) && same_bits(ws.n, ps.n) && ws.t == ps.t;
            // the offsets the slabs report add up to the whole
            const std::vector<mc_shard>& sh = parts.shards();
            bool ok_o = (int)sh.size() == n && sh[0].tri_offset == 0 && sh[(size_t)n - 1].n_tris_total == ps.t.size() / 3 && sh[0].z_begin == 0;
            for (int i = 1; ok_o && i < n; ++i) ok_o = sh[(size_t)i].z_begin == sh[(size_t)i - 1].z_end && sh[(size_t)i].tri_offset >= sh[(size_t)i - 1].tri_offset;
            std::printf("%s step %g x%d: verts=%zu tris=%zu indexed %s soup %s offsets %s\n", c.eq, (double)c.step, n, wi.v.size() / 3, wi.t.size() / 3,
                        ok_i ? "same" : "DIFFERENT", ok_s ? "same" : "DIFFERENT", ok_o ? "ok" : "BAD");
            if (!(ok_i && ok_s && ok_o)) ++bad;
            if (wi.t.empty() && std::strcmp(c.eq, "z") != 0) ++bad;  // (a case that compares nothing proves nothing)
        }
        // seed mode follows one component through the whole grid: with a device list set the first device sweeps everything,
        // and the mesh is the one a Marching object without a list gets
        if (!*c.cons && c.sx == 1.0f) {
            Mesh s1, s2;
            Marching seeded, seeded_list;
            for (Marching* m : {&seeded, &seeded_list}) {
                m->set_evaluator(&ev);
                m->set_grid_step_size(c.step);
                m->set_surface_constant(c.iso);
                m->seed_mode(true);
                m->set_seed(0.0f, 0.0f, 1.0f);
            }
            seeded_list.set_devices({base.empty() ? 0 : base[0], base.empty() ? 0 : base[0], base.empty() ? 0 : base[0]});
            if (!sweep(seeded, true, s1) || !sweep(seeded_list, true, s2)) return 7;
            const bool ok = same_bits(s1.v, s2.v) && s1.t == s2.t && same_bits(s1.n, s2.n);
            std::printf("%s seed mode with a device list: tris=%zu %s\n", c.eq, s2.t.size() / 3, ok ? "same" : "DIFFERENT");
            if (!ok) ++bad;
        }
    }
    if (bad) {
        std::printf("SHARDED_FAILED %d\n", bad);
        return 1;
    }
    std::printf("SHARDED_OK\n");
    return 0;
}
