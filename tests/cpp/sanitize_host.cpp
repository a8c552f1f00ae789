// tests/cpp/sanitize_host.cpp — the product's host-only logic under AddressSanitizer + UBSan (CPU build; GPU sanitizers
// are not available on the pool).  Compiled from the product sources directly: csrc/bge_flatten.cpp (flattening and
// subtree partition) and host/bge/scene_json.hpp (the reference scene-format parser), driven with random and
// adversarial inputs.  Any heap overflow, use-after-free, signed overflow or misaligned access aborts the program.
#include <cstdint>
#include <cstdio>
#include <random>
#include <string>
#include <vector>

#include "../../banggameengine_amd/csrc/bge_flatten.hpp"
#include "../../banggameengine_amd/host/bge/scene.hpp"
#include "../../banggameengine_amd/host/bge/scene_json.hpp"

namespace {

int failures = 0;
#define CHECK(cond)                                                      \
    do {                                                                 \
        if (!(cond)) {                                                   \
            std::fprintf(stderr, "%s:%d: CHECK(%s)\n", __FILE__, __LINE__, #cond); \
            ++failures;                                                  \
        }                                                                \
    } while (0)

void check_layout(uint64_t n, const std::vector<uint32_t>& parent, const std::vector<uint8_t>* has_tf)
{
    bge::Flattened f;
    bge::flatten_topology(n, parent.empty() ? nullptr : parent.data(), has_tf ? has_tf->data() : nullptr, f);
    CHECK(f.n_entities == n);
    CHECK(f.slot_of_entity.size() == n);
    CHECK(f.n_slots == static_cast<uint64_t>(f.n_tiles_total) * bge::kTile);
    CHECK(f.entity_of_slot.size() == f.n_slots && f.flags.size() == f.n_slots && f.parent_field.size() == f.n_slots);
    uint64_t placed = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const uint32_t s = f.slot_of_entity[i];
        const bool tf = !has_tf || (*has_tf)[i];
        if (!tf) {
            CHECK(s == bge::kNone);
            continue;
        }
        CHECK(s != bge::kNone && s < f.n_slots);
        if (s != bge::kNone && s < f.n_slots) {
            CHECK(f.entity_of_slot[s] == i);
            CHECK(f.flags[s] & bge::kValid);
            ++placed;
        }
    }
    CHECK(placed == f.n_transforms);
    for (size_t p = 0; p + 1 < f.pass_tile_begin.size(); ++p) CHECK(f.pass_tile_begin[p] <= f.pass_tile_begin[p + 1]);
    if (!f.pass_tile_begin.empty()) CHECK(f.pass_tile_begin.back() == f.n_tiles_ticked);
    for (uint32_t nr : {1u, 2u, 3u, 8u}) {
        std::vector<uint32_t> rank(n);
        std::vector<uint64_t> load(nr);
        bge::partition_subtrees(n, parent.empty() ? nullptr : parent.data(), has_tf ? has_tf->data() : nullptr, nr, rank.data(), load.data());
        uint64_t total = 0;
        for (uint64_t l : load) total += l;
        CHECK(total <= n);
        for (uint64_t i = 0; i < n; ++i) CHECK(rank[i] < nr);
    }
}

void fuzz_flatten()
{
    std::mt19937 rng(12345);
    for (int round = 0; round < 300; ++round) {
        const uint64_t n = rng() % 2000;
        std::vector<uint32_t> parent(n, bge::kNone);
        const int mode = round % 6;
        for (uint64_t i = 0; i < n; ++i) {
            switch (mode) {
            case 0: if (i && rng() % 10) parent[i] = rng() % i; break;                 // random forest
            case 1: if (i) parent[i] = static_cast<uint32_t>(i - 1); break;            // one deep chain
            case 2: if (i) parent[i] = 0; break;                                       // one wide root
            case 3: parent[i] = rng() % (n + 5); break;                                // arbitrary: cycles, self loops, out of range
            case 4: if (i % 64) parent[i] = static_cast<uint32_t>(i - 1 - (rng() % (i % 64))); break; // small subtrees
            default: break;                                                            // flat
            }
        }
        std::vector<uint8_t> has(n);
        for (auto& h : has) h = rng() % 8 != 0;
        check_layout(n, parent, nullptr);
        check_layout(n, parent, &has);
    }
    check_layout(0, {}, nullptr);
    check_layout(70000, std::vector<uint32_t>(70000, bge::kNone), nullptr); // the flat fast path
}

void fuzz_json()
{
    const std::string good = R"({"entities":[{"id":1,"name":"a","transform":{"position":[1,2,3],"rotationEuler":[0,0.5,0],"scale":[1,1,1]},
        "rigidBody":{"type":"dynamic","mass":2.0},"collider":{"shape":"box","size":[0.5,0.5,0.5]},"children":[2]},
        {"id":2,"transform":{"position":[0,1,0]},"parent":1}]})";
    bge::Scene scene;
    std::string err;
    CHECK(bge::LoadSceneFromJsonText(good, scene, &err));
    std::mt19937 rng(777);
    // truncations, single-byte corruptions, deep nesting: must fail cleanly or succeed, never touch bad memory
    for (size_t cut = 0; cut < good.size(); cut += 3) {
        bge::Scene s;
        (void)bge::LoadSceneFromJsonText(good.substr(0, cut), s, &err);
    }
    for (int k = 0; k < 3000; ++k) {
        std::string t = good;
        const int edits = 1 + rng() % 4;
        for (int e = 0; e < edits; ++e) t[rng() % t.size()] = static_cast<char>(rng() % 256);
        bge::Scene s;
        (void)bge::LoadSceneFromJsonText(t, s, &err);
    }
    for (int depth : {10, 1000, 100000}) {
        std::string t(static_cast<size_t>(depth), '[');
        bge::Scene s;
        (void)bge::LoadSceneFromJsonText(t, s, &err);
        std::string u = "{\"entities\":" + std::string(static_cast<size_t>(depth), '[') + std::string(static_cast<size_t>(depth), ']') + "}";
        (void)bge::LoadSceneFromJsonText(u, s, &err);
    }
    const char* odd[] = {"", "{", "}", "null", "{\"entities\":null}", "{\"entities\":[{}]}", "{\"entities\":[{\"id\":-1}]}",
                         "{\"entities\":[{\"id\":1e400}]}", "{\"entities\":[{\"id\":1,\"parent\":1}]}",
                         "{\"entities\":[{\"id\":1,\"transform\":{\"position\":[1]}}]}", "\xff\xfe\x00", "{\"a\":\"\\u12\"}"};
    for (const char* o : odd) {
        bge::Scene s;
        (void)bge::LoadSceneFromJsonText(o, s, &err);
    }
}

} // namespace

int main()
{
    fuzz_flatten();
    fuzz_json();
    if (failures) {
        std::fprintf(stderr, "%d checks failed\n", failures);
        return 1;
    }
    std::puts("sanitize_host ok");
    return 0;
}
