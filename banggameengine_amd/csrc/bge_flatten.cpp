// bge_flatten.cpp — see bge_flatten.hpp.
#include "bge_flatten.hpp"

#include <algorithm>
#include <numeric>

namespace bge {

namespace {

struct Graph {
    uint64_t n = 0;
    std::vector<uint32_t> eff_parent;  // parent that owns a Transform, else kNone
    std::vector<uint8_t> has_tf;
    std::vector<uint32_t> child_begin; // CSR over eff_parent
    std::vector<uint32_t> child_list;
    std::vector<uint32_t> bfs;         // every node reachable from a root, parents before children
    std::vector<uint32_t> subtree;     // node count of the subtree rooted here (reachable nodes only)
    std::vector<uint32_t> depth;       // global depth, roots = 0 (kNone if unreachable)
    std::vector<uint32_t> roots;       // in entity order
};

void build_graph(uint64_t n, const uint32_t* parent, const uint8_t* has_transform, Graph& g)
{
    g.n = n;
    g.has_tf.assign(n, 1);
    if (has_transform) {
        for (uint64_t i = 0; i < n; ++i) g.has_tf[i] = has_transform[i] ? 1 : 0;
    }
    g.eff_parent.assign(n, kNone);
    g.child_begin.assign(n + 1, 0);
    for (uint64_t i = 0; i < n; ++i) {
        if (!g.has_tf[i]) continue;
        const uint32_t p = parent ? parent[i] : kNone;
        // Scene.cpp:528 — a parent without a Transform makes the child a root
        if (p != kNone && p < n && g.has_tf[p]) {
            g.eff_parent[i] = p;
            g.child_begin[p + 1]++;
        }
    }
    for (uint64_t i = 0; i < n; ++i) g.child_begin[i + 1] += g.child_begin[i];
    g.child_list.resize(g.child_begin[n]);
    {
        std::vector<uint32_t> cursor(g.child_begin.begin(), g.child_begin.end() - 1);
        for (uint64_t i = 0; i < n; ++i) {
            const uint32_t p = g.eff_parent[i];
            if (p != kNone) g.child_list[cursor[p]++] = static_cast<uint32_t>(i);
        }
    }
    g.depth.assign(n, kNone);
    g.bfs.clear();
    g.bfs.reserve(n);
    g.roots.clear();
    for (uint64_t i = 0; i < n; ++i) {
        if (g.has_tf[i] && g.eff_parent[i] == kNone) {
            g.roots.push_back(static_cast<uint32_t>(i));
            g.depth[i] = 0;
            g.bfs.push_back(static_cast<uint32_t>(i));
        }
    }
    for (size_t head = 0; head < g.bfs.size(); ++head) {
        const uint32_t u = g.bfs[head];
        for (uint32_t c = g.child_begin[u]; c < g.child_begin[u + 1]; ++c) {
            const uint32_t v = g.child_list[c];
            g.depth[v] = g.depth[u] + 1;
            g.bfs.push_back(v);
        }
    }
    g.subtree.assign(n, 0);
    for (size_t k = g.bfs.size(); k-- > 0;) {
        const uint32_t u = g.bfs[k];
        g.subtree[u] += 1;
        if (g.eff_parent[u] != kNone) g.subtree[g.eff_parent[u]] += g.subtree[u];
    }
}

struct TileBuilder {
    // nodes of the tile being filled, with their in-tile level
    std::vector<uint32_t> nodes;
    std::vector<uint8_t> levels;
    void clear()
    {
        nodes.clear();
        levels.clear();
    }
    uint32_t room() const { return kTile - static_cast<uint32_t>(nodes.size()); }
};

} // namespace

void flatten_topology(uint64_t n, const uint32_t* parent, const uint8_t* has_transform, Flattened& out)
{
    Graph g;
    build_graph(n, parent, has_transform, g);

    out = Flattened{};
    out.n_entities = n;
    out.slot_of_entity.assign(n, kNone);
    out.pass_of_entity.assign(n, kNone);
    out.pass_tile_begin.assign(1, 0);

    std::vector<uint32_t> tile_nodes_all;  // concatenated slots -> entity (kNone padding)
    std::vector<uint32_t> in_tile_index(n, kNone);
    std::vector<uint32_t> tile_of_node(n, kNone);

    auto emit_tile = [&](TileBuilder& tb, bool limbo) {
        if (tb.nodes.empty()) return;
        const uint32_t tile = static_cast<uint32_t>(out.tile_hdr.size());
        const uint32_t count = static_cast<uint32_t>(tb.nodes.size());
        // stable counting sort by level
        uint32_t max_level = 0;
        for (uint8_t l : tb.levels) max_level = std::max<uint32_t>(max_level, l);
        std::vector<uint32_t> start(max_level + 2, 0);
        for (uint8_t l : tb.levels) start[l + 1]++;
        for (uint32_t l = 0; l <= max_level; ++l) start[l + 1] += start[l];
        std::vector<uint32_t> ordered(count);
        std::vector<uint8_t> ordered_level(count);
        for (uint32_t k = 0; k < count; ++k) {
            const uint32_t dst = start[tb.levels[k]]++;
            ordered[dst] = tb.nodes[k];
            ordered_level[dst] = tb.levels[k];
        }
        const uint64_t base = static_cast<uint64_t>(tile) * kTile;
        out.entity_of_slot.resize(base + kTile, kNone);
        out.parent_field.resize(base + kTile, kNone);
        out.flags.resize(base + kTile, 0);
        bool any_ext = false;
        for (uint32_t k = 0; k < count; ++k) {
            const uint32_t e = ordered[k];
            in_tile_index[e] = k;
            tile_of_node[e] = tile;
            out.slot_of_entity[e] = static_cast<uint32_t>(base + k);
            out.entity_of_slot[base + k] = e;
        }
        for (uint32_t k = 0; k < count; ++k) {
            const uint32_t e = ordered[k];
            uint32_t f = kValid | (static_cast<uint32_t>(ordered_level[k]) << kLevelShift);
            if (limbo) {
                f |= kLimbo;
            } else {
                const uint32_t p = g.eff_parent[e];
                if (p != kNone) {
                    f |= kHasParent;
                    if (tile_of_node[p] == tile) {
                        out.parent_field[base + k] = in_tile_index[p];
                    } else {
                        f |= kExtParent;
                        any_ext = true;
                        out.parent_field[base + k] = out.slot_of_entity[p]; // placed in an earlier pass
                    }
                }
            }
            out.flags[base + k] = f;
        }
        uint32_t hdr = (max_level & kHdrLevelMask) | (count << kHdrCountShift);
        if (any_ext) hdr |= kHdrExt;
        out.tile_hdr.push_back(hdr);
        tb.clear();
    };

    // ---- passes over pending subtrees
    std::vector<uint32_t> pending = g.roots;
    std::vector<uint32_t> next_pending;
    std::vector<uint32_t> queue;
    std::vector<uint8_t> queue_level;
    TileBuilder tb;
    uint32_t pass = 0;
    while (!pending.empty()) {
        next_pending.clear();
        for (uint32_t r : pending) {
            const uint32_t size = g.subtree[r];
            const bool whole = size <= kTile;
            if (whole && size > tb.room()) emit_tile(tb, false);
            if (!whole && !tb.nodes.empty()) emit_tile(tb, false);
            // breadth-first over the subtree; an oversize subtree is cut when the tile is full
            queue.clear();
            queue_level.clear();
            queue.push_back(r);
            queue_level.push_back(0);
            for (size_t head = 0; head < queue.size(); ++head) {
                const uint32_t u = queue[head];
                const uint8_t lvl = queue_level[head];
                if (tb.room() == 0) {
                    // cut: u (whose parent is already placed) starts a subtree of the next pass
                    next_pending.push_back(u);
                    continue;
                }
                tb.nodes.push_back(u);
                tb.levels.push_back(lvl);
                out.pass_of_entity[u] = pass;
                for (uint32_t c = g.child_begin[u]; c < g.child_begin[u + 1]; ++c) {
                    queue.push_back(g.child_list[c]);
                    queue_level.push_back(static_cast<uint8_t>(lvl + 1));
                }
            }
            if (!whole) emit_tile(tb, false);
        }
        emit_tile(tb, false);
        out.pass_tile_begin.push_back(static_cast<uint32_t>(out.tile_hdr.size()));
        pending.swap(next_pending);
        ++pass;
    }
    out.n_tiles_ticked = static_cast<uint32_t>(out.tile_hdr.size());

    // ---- limbo: Transform-bearing entities never reached from a root (parent cycles)
    for (uint64_t i = 0; i < n; ++i) {
        if (g.has_tf[i] && g.depth[i] == kNone) {
            if (tb.room() == 0) emit_tile(tb, true);
            tb.nodes.push_back(static_cast<uint32_t>(i));
            tb.levels.push_back(0);
            out.n_limbo++;
        }
    }
    emit_tile(tb, true);
    out.n_tiles_total = static_cast<uint32_t>(out.tile_hdr.size());
    out.n_slots = static_cast<uint64_t>(out.n_tiles_total) * kTile;
    out.entity_of_slot.resize(out.n_slots, kNone);
    out.parent_field.resize(out.n_slots, kNone);
    out.flags.resize(out.n_slots, 0);

    out.n_transforms = g.bfs.size() + out.n_limbo;
    out.root_slots.reserve(g.roots.size());
    for (uint32_t r : g.roots) out.root_slots.push_back(out.slot_of_entity[r]);
    for (uint32_t u : g.bfs) out.max_depth = std::max(out.max_depth, g.depth[u]);
}

void partition_subtrees(uint64_t n, const uint32_t* parent, const uint8_t* has_transform, uint32_t nranks,
                        uint32_t* rank_of_entity, uint64_t* nodes_per_rank)
{
    Graph g;
    build_graph(n, parent, has_transform, g);
    std::vector<uint64_t> load(nranks, 0);
    for (uint64_t i = 0; i < n; ++i) rank_of_entity[i] = 0;

    // Flat scenes (every root a singleton) degenerate to contiguous equal ranges, which also keeps
    // each rank's entities contiguous in memory.
    const bool all_singletons = g.bfs.size() == g.roots.size();
    if (all_singletons) {
        const uint64_t m = g.roots.size();
        for (uint64_t k = 0; k < m; ++k) {
            const uint32_t r = static_cast<uint32_t>((k * nranks) / (m ? m : 1));
            rank_of_entity[g.roots[k]] = r;
            load[r]++;
        }
    } else {
        // largest subtree first onto the least-loaded rank; ties broken by entity order for determinism
        std::vector<uint32_t> order(g.roots.size());
        std::iota(order.begin(), order.end(), 0u);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
            return g.subtree[g.roots[a]] > g.subtree[g.roots[b]];
        });
        std::vector<uint32_t> rank_of_root(g.roots.size(), 0);
        // equal-size subtrees: round-robin over contiguous blocks keeps shards contiguous
        for (uint32_t k : order) {
            uint32_t best = 0;
            for (uint32_t r = 1; r < nranks; ++r) {
                if (load[r] < load[best]) best = r;
            }
            rank_of_root[k] = best;
            load[best] += g.subtree[g.roots[k]];
        }
        // propagate to descendants in BFS order (parents first)
        std::vector<uint32_t> root_index_of(n, kNone);
        for (uint32_t k = 0; k < g.roots.size(); ++k) rank_of_entity[g.roots[k]] = rank_of_root[k];
        for (uint32_t u : g.bfs) {
            if (g.eff_parent[u] != kNone) rank_of_entity[u] = rank_of_entity[g.eff_parent[u]];
        }
    }
    if (nodes_per_rank) {
        for (uint32_t r = 0; r < nranks; ++r) nodes_per_rank[r] = load[r];
    }
}

} // namespace bge
