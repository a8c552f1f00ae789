// bge_flatten.cpp — see bge_flatten.hpp.
#include "bge_flatten.hpp"

#include <algorithm>
#include <cstdlib>
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
    std::vector<uint32_t> height;      // levels below this node (leaf = 0)
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
    g.height.assign(n, 0);
    for (size_t k = g.bfs.size(); k-- > 0;) {
        const uint32_t u = g.bfs[k];
        g.subtree[u] += 1;
        const uint32_t p = g.eff_parent[u];
        if (p != kNone) {
            g.subtree[p] += g.subtree[u];
            g.height[p] = std::max(g.height[p], g.height[u] + 1);
        }
    }
}

struct TileBuilder {
    // Nodes of the tile being filled with their in-tile level.  A wave-local tile keeps four groups (one per
    // wave64) and only ever receives whole subtrees of <= 64 nodes; a block tile uses group 0 as a flat list.
    std::vector<uint32_t> nodes[4];
    std::vector<uint8_t> levels[4];
    bool wave_local = false;
    void clear()
    {
        for (int g = 0; g < 4; ++g) {
            nodes[g].clear();
            levels[g].clear();
        }
    }
    bool empty() const { return nodes[0].empty() && nodes[1].empty() && nodes[2].empty() && nodes[3].empty(); }
    uint32_t size() const { return static_cast<uint32_t>(nodes[0].size() + nodes[1].size() + nodes[2].size() + nodes[3].size()); }
    uint32_t room() const { return kTile - size(); } // block tiles
    int group_with_room(uint32_t need) const
    {
        for (int g = 0; g < 4; ++g) {
            if (kGroup - nodes[g].size() >= need) return g;
        }
        return -1;
    }
    void reserve()
    {
        for (int g = 0; g < 4; ++g) {
            nodes[g].reserve(kTile);
            levels[g].reserve(kTile);
        }
    }
};

} // namespace

FlattenOptions flatten_options_from_env()
{
    FlattenOptions o;
    if (const char* s = std::getenv("BGE_WAVE_LOCAL_HEIGHT")) o.wave_local_max_height = static_cast<uint32_t>(std::atoi(s));
    return o;
}

void flatten_topology(uint64_t n, const uint32_t* parent, const uint8_t* has_transform, Flattened& out,
                      const FlattenOptions& opt)
{
    // Flat scene (no parent links at all, every entity owns a Transform): slot == entity index, wave-local tiles,
    // no graph needed.  (16 M entities: 0.6 s instead of 5-7 s, which was first-touch of ~1.2 GB of temporaries.)
    bool flat = has_transform == nullptr;
    if (flat && parent) {
        for (uint64_t i = 0; i < n && flat; ++i) flat = parent[i] == kNone || parent[i] >= n;
    }
    if (flat) {
        out = Flattened{};
        out.n_entities = out.n_transforms = n;
        out.identity = true;
        out.n_tiles_ticked = out.n_tiles_total = static_cast<uint32_t>((n + kTile - 1) / kTile);
        out.n_slots = static_cast<uint64_t>(out.n_tiles_total) * kTile;
        out.slot_of_entity.resize(n);
        out.root_slots.resize(n);
        out.entity_of_slot.assign(out.n_slots, kNone);
        out.parent_field.assign(out.n_slots, kNone);
        out.flags.assign(out.n_slots, 0);
        out.pass_of_entity.assign(n, 0);
        for (uint64_t i = 0; i < n; ++i) {
            out.slot_of_entity[i] = out.root_slots[i] = out.entity_of_slot[i] = static_cast<uint32_t>(i);
            out.flags[i] = kValid;
        }
        out.tile_hdr.assign(out.n_tiles_total, (kTile << kHdrCountShift) | kHdrWaveLocal);
        if (n % kTile) out.tile_hdr.back() = (static_cast<uint32_t>(n % kTile) << kHdrCountShift) | kHdrWaveLocal;
        out.pass_tile_begin.assign(1, 0);
        if (out.n_tiles_total) out.pass_tile_begin.push_back(out.n_tiles_total);
        return;
    }

    Graph g;
    build_graph(n, parent, has_transform, g);

    out = Flattened{};
    out.n_entities = n;
    out.slot_of_entity.assign(n, kNone);
    out.pass_of_entity.assign(n, kNone);
    out.pass_tile_begin.assign(1, 0);

    std::vector<uint32_t> in_tile_index(n, kNone);
    std::vector<uint32_t> tile_of_node(n, kNone);

    // stable counting sort of one node list by level; returns the highest level
    auto sort_by_level = [](const std::vector<uint32_t>& nodes, const std::vector<uint8_t>& levels,
                            std::vector<uint32_t>& out_nodes, std::vector<uint8_t>& out_levels) {
        uint32_t max_level = 0;
        for (uint8_t l : levels) max_level = std::max<uint32_t>(max_level, l);
        std::vector<uint32_t> start(max_level + 2, 0);
        for (uint8_t l : levels) start[l + 1]++;
        for (uint32_t l = 0; l <= max_level; ++l) start[l + 1] += start[l];
        out_nodes.resize(nodes.size());
        out_levels.resize(nodes.size());
        for (size_t k = 0; k < nodes.size(); ++k) {
            const uint32_t dst = start[levels[k]]++;
            out_nodes[dst] = nodes[k];
            out_levels[dst] = levels[k];
        }
        return max_level;
    };

    auto emit_tile = [&](TileBuilder& tb, bool limbo) {
        if (tb.empty()) return;
        const uint32_t tile = static_cast<uint32_t>(out.tile_hdr.size());
        const uint64_t base = static_cast<uint64_t>(tile) * kTile;
        out.entity_of_slot.resize(base + kTile, kNone);
        out.parent_field.resize(base + kTile, kNone);
        out.flags.resize(base + kTile, 0);

        // (in-tile position, entity, level) of every node of the tile
        std::vector<uint32_t> place, ent;
        std::vector<uint8_t> lvl;
        uint32_t max_level = 0;
        std::vector<uint32_t> sn;
        std::vector<uint8_t> sl;
        const int n_groups = tb.wave_local ? 4 : 1;
        for (int grp = 0; grp < n_groups; ++grp) {
            if (tb.nodes[grp].empty()) continue;
            max_level = std::max(max_level, sort_by_level(tb.nodes[grp], tb.levels[grp], sn, sl));
            for (size_t k = 0; k < sn.size(); ++k) {
                place.push_back(static_cast<uint32_t>(grp * kGroup + k)); // block tiles: grp == 0, a prefix of the tile
                ent.push_back(sn[k]);
                lvl.push_back(sl[k]);
            }
        }
        const uint32_t count = static_cast<uint32_t>(ent.size());
        for (uint32_t k = 0; k < count; ++k) {
            const uint32_t e = ent[k];
            in_tile_index[e] = place[k];
            tile_of_node[e] = tile;
            out.slot_of_entity[e] = static_cast<uint32_t>(base + place[k]);
            out.entity_of_slot[base + place[k]] = e;
        }
        bool any_ext = false;
        for (uint32_t k = 0; k < count; ++k) {
            const uint32_t e = ent[k];
            const uint64_t s = base + place[k];
            uint32_t f = kValid | (static_cast<uint32_t>(lvl[k]) << kLevelShift);
            if (!limbo) {
                const uint32_t p = g.eff_parent[e];
                if (p != kNone) {
                    f |= kHasParent;
                    if (tile_of_node[p] == tile) {
                        out.parent_field[s] = in_tile_index[p];
                        f |= in_tile_index[p] << kParentShift; // the kernel reads it from the flag word
                    } else {
                        f |= kExtParent;
                        any_ext = true;
                        out.parent_field[s] = out.slot_of_entity[p]; // placed in an earlier pass
                    }
                }
            }
            out.flags[s] = f;
        }
        uint32_t hdr = (max_level & kHdrLevelMask) | (count << kHdrCountShift);
        if (any_ext) hdr |= kHdrExt;
        if (tb.wave_local) hdr |= kHdrWaveLocal;
        out.tile_hdr.push_back(hdr);
        tb.clear();
    };

    // ---- passes over pending subtrees
    std::vector<uint32_t> pending = g.roots;
    std::vector<uint32_t> next_pending;
    std::vector<uint32_t> queue;
    std::vector<uint8_t> queue_level;
    TileBuilder wl, bk; // the open wave-local tile and the open block tile
    wl.wave_local = true;
    bk.wave_local = false;
    wl.reserve();
    bk.reserve();
    out.tile_hdr.reserve(n / kTile + 8);
    out.entity_of_slot.reserve(n + 2 * kTile);
    out.parent_field.reserve(n + 2 * kTile);
    out.flags.reserve(n + 2 * kTile);
    uint32_t pass = 0;
    while (!pending.empty()) {
        next_pending.clear();
        for (uint32_t r : pending) {
            const uint32_t size = g.subtree[r];
            if (size == 1) {
                // singleton (every entity of a flat scene): no walk needed
                int grp = wl.group_with_room(1);
                if (grp < 0) {
                    emit_tile(wl, false);
                    grp = 0;
                }
                wl.nodes[grp].push_back(r);
                wl.levels[grp].push_back(0);
                out.pass_of_entity[r] = pass;
                continue;
            }
            // destination of the subtree's nodes
            std::vector<uint32_t>* dst_nodes;
            std::vector<uint8_t>* dst_levels;
            uint32_t budget; // nodes that may still be placed (oversize subtrees are cut when it reaches 0)
            if (size <= kGroup && g.height[r] <= opt.wave_local_max_height) {
                int grp = wl.group_with_room(size);
                if (grp < 0) {
                    emit_tile(wl, false);
                    grp = 0;
                }
                dst_nodes = &wl.nodes[grp];
                dst_levels = &wl.levels[grp];
                budget = size;
            } else {
                const bool whole = size <= kTile;
                if (whole ? size > bk.room() : !bk.empty()) emit_tile(bk, false);
                dst_nodes = &bk.nodes[0];
                dst_levels = &bk.levels[0];
                budget = whole ? size : kTile;
            }
            // breadth-first over the subtree; an oversize subtree is cut when its tile is full
            queue.clear();
            queue_level.clear();
            queue.push_back(r);
            queue_level.push_back(0);
            for (size_t head = 0; head < queue.size(); ++head) {
                const uint32_t u = queue[head];
                const uint8_t lvl = queue_level[head];
                if (budget == 0) {
                    // cut: u (whose parent is already placed) starts a subtree of the next pass
                    next_pending.push_back(u);
                    continue;
                }
                --budget;
                dst_nodes->push_back(u);
                dst_levels->push_back(lvl);
                out.pass_of_entity[u] = pass;
                for (uint32_t ch = g.child_begin[u]; ch < g.child_begin[u + 1]; ++ch) {
                    queue.push_back(g.child_list[ch]);
                    queue_level.push_back(static_cast<uint8_t>(lvl + 1));
                }
            }
            if (size > kTile) emit_tile(bk, false);
        }
        emit_tile(wl, false);
        emit_tile(bk, false);
        out.pass_tile_begin.push_back(static_cast<uint32_t>(out.tile_hdr.size()));
        pending.swap(next_pending);
        ++pass;
    }
    out.n_tiles_ticked = static_cast<uint32_t>(out.tile_hdr.size());

    // ---- limbo: Transform-bearing entities never reached from a root (parent cycles)
    TileBuilder lb;
    lb.wave_local = false;
    for (uint64_t i = 0; i < n; ++i) {
        if (g.has_tf[i] && g.depth[i] == kNone) {
            if (lb.room() == 0) emit_tile(lb, true);
            lb.nodes[0].push_back(static_cast<uint32_t>(i));
            lb.levels[0].push_back(0);
            out.n_limbo++;
        }
    }
    emit_tile(lb, true);
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
