// oracle/soa_ref.h — TEST INFRASTRUCTURE ONLY (see oracle/README.md).
//
// "CPU-opt" baseline of BASELINE.md §3: the same arithmetic as ecs_ref.h / physics_ref.h (bx mtxSRT / mtxMul,
// free-body semi-implicit Euler for non-spinning Dynamic bodies) on dense structure-of-arrays, nodes grouped by depth,
// all host threads (OpenMP).  It is what a competent CPU rewrite of the reference's hot path would look like, so that
// the GPU numbers can also be read hardware-against-hardware; it is NOT how the reference works (hash maps, recursion,
// one thread) — that is ecs_ref.h, the "port".  Results are bit-identical to the port (tests/test_oracle_golden.py).
#pragma once

#include <cstdint>
#include <cstring>
#include <vector>

#include "bx_math.h"

namespace orc {

struct SoaScene {
    uint64_t n = 0;
    std::vector<int32_t> parent;        // -1 = root
    std::vector<float> pos, euler, scale, vel; // n x 3
    std::vector<uint8_t> dynamic;       // 1 = Dynamic body with mass 1
    std::vector<float> local, world;    // n x 16
    std::vector<uint32_t> order;        // nodes sorted by depth
    std::vector<uint64_t> level_begin;  // offsets into order per depth

    void build_levels()
    {
        std::vector<uint32_t> depth(n, 0);
        uint32_t max_depth = 0;
        for (uint64_t i = 0; i < n; ++i) { // parents precede children in the synthetic workloads
            if (parent[i] >= 0) depth[i] = depth[parent[i]] + 1;
            max_depth = depth[i] > max_depth ? depth[i] : max_depth;
        }
        level_begin.assign(max_depth + 2, 0);
        for (uint64_t i = 0; i < n; ++i) level_begin[depth[i] + 1]++;
        for (uint32_t d = 0; d <= max_depth; ++d) level_begin[d + 1] += level_begin[d];
        order.resize(n);
        std::vector<uint64_t> cur(level_begin.begin(), level_begin.end() - 1);
        for (uint64_t i = 0; i < n; ++i) order[cur[depth[i]]++] = static_cast<uint32_t>(i);
    }

    void tick(float dt, float gy)
    {
        const int64_t N = static_cast<int64_t>(n);
        const float imp = ((gy * 1.0f) * 1.0f) * dt; // F = g*m, v += (F*inv_m)*dt with m = 1
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < N; ++i) {
            float* p = &pos[3 * i];
            if (dynamic[i]) {
                float* v = &vel[3 * i];
                v[0] = v[0] + ((0.0f * 1.0f) * 1.0f) * dt;
                v[1] = v[1] + imp;
                v[2] = v[2] + ((0.0f * 1.0f) * 1.0f) * dt;
                p[0] = p[0] + v[0] * dt;
                p[1] = p[1] + v[1] * dt;
                p[2] = p[2] + v[2] * dt;
            }
            const float* e = &euler[3 * i];
            const float* s = &scale[3 * i];
            bxm::mtxSRT(&local[16 * i], s[0], s[1], s[2], e[0], e[1], e[2], p[0], p[1], p[2]);
            if (parent[i] < 0) std::memcpy(&world[16 * i], &local[16 * i], 64);
        }
        for (size_t d = 1; d + 1 < level_begin.size(); ++d) {
            const int64_t b = static_cast<int64_t>(level_begin[d]), e = static_cast<int64_t>(level_begin[d + 1]);
#pragma omp parallel for schedule(static)
            for (int64_t k = b; k < e; ++k) {
                const uint32_t i = order[k];
                bxm::mtxMul(&world[16ull * i], &world[16ull * parent[i]], &local[16ull * i]);
            }
        }
    }
};

} // namespace orc
