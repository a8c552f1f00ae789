// bge_flatten.hpp — host-side flattening of the scene graph into depth-ordered 256-slot tiles.
//
// Replaces the pointer-chasing structures the reference walks every tick
// (Scene::m_parents / m_children, src/ecs/Scene.h:104-105; the root scan of
// Scene::ForEachRootTransform, src/ecs/Scene.cpp:523-533; the recursion of UpdateNode,
// src/ecs/TransformSystem.cpp:10-37) by a layout computed once per topology change:
//
//   * every Transform-bearing entity gets a SLOT; slots are grouped in TILES of 256 (one workgroup);
//   * a tile holds whole subtrees, its nodes sorted by their depth inside the tile ("level"), so the
//     kernel resolves level d after level d-1 with the parent's world matrix staged in LDS;
//   * shallow subtrees of at most 64 nodes (FlattenOptions; by default only singletons, i.e. flat scenes) are
//     packed first-fit into the tile's four 64-slot GROUPS, one group per wave64: such a "wave-local" tile needs
//     no workgroup barrier at all — each wave resolves and writes out its own subtrees.  Everything else shares
//     level-major "block" tiles (one barrier per level, every level step runs with full waves);
//   * a subtree larger than a tile is cut breadth-first; the cut-off children are placed in a later
//     PASS (a dependent launch) and read their parent's world matrix from global memory;
//   * entities in a parent cycle are unreachable from any root (the reference never updates them,
//     SURVEY.md App. B.3): they get storage in trailing "limbo" tiles that are never ticked.
#pragma once

#include <cstdint>
#include <vector>

namespace bge {

constexpr uint32_t kTile = 256;
constexpr uint32_t kNone = 0xffffffffu;
constexpr uint32_t kFrozenGhost = 0xfffffffeu; // TriggerView::slot: the entity lost its Transform, the ghost keeps the box it has

// per-slot flag word (also the device encoding)
constexpr uint32_t kTypeMask = 0x3u;     // 0 none, 1 Static, 2 Dynamic, 3 Kinematic  (= bge_body_type + 1)
constexpr uint32_t kTDirty = 0x4u;       // Transform.dirty
constexpr uint32_t kBDirty = 0x8u;       // RigidBody.dirty or Collider.dirty
constexpr uint32_t kSpin = 0x10u;        // angular velocity != 0
constexpr uint32_t kValid = 0x20u;       // slot holds a Transform
constexpr uint32_t kHasParent = 0x40u;
constexpr uint32_t kExtParent = 0x80u;   // parent field is a global slot resolved in an earlier pass
constexpr uint32_t kLevelShift = 8;      // bits 8..15: level inside the tile
constexpr uint32_t kLevelMask = 0xff00u;
constexpr uint32_t kParentShift = 16;    // bits 16..23: in-tile index of the parent (when kHasParent && !kExtParent)
constexpr uint32_t kParentMask = 0xff0000u;
constexpr uint32_t kDrowsy = 1u << 24;   // the body's deactivation record is non-zero (timer running, wants to sleep, asleep)
constexpr uint32_t kSettled = 1u << 25;  // (BGE_TICK_BULLET_BASIS) the stored quaternion is a fixed point of Bullet's orientation round trip at zero angular
                                         // velocity and rotationEuler holds its angles: the step leaves both alone.  Cleared by whatever writes the quaternion.
constexpr uint32_t kMassShift = 26;      // bits 26..31: mass class (index into the world's mass palette);
constexpr uint32_t kMassMask = 0xfc000000u; //            63 = read the per-slot inv_mass array instead
constexpr uint32_t kMassClassArray = 63;

// Deactivation record (one 32-bit word per slot, read only while kDrowsy is set): the bits of btCollisionObject's
// m_deactivationTime (a float >= 0) while the body is ACTIVE_TAG, or one of two negative-NaN sentinels.
constexpr uint32_t kDeactSleeping = 0xffc00002u; // ISLAND_SLEEPING
constexpr uint32_t kDeactWants = 0xffc00003u;    // WANTS_DEACTIVATION

// per-tile header word
constexpr uint32_t kHdrLevelMask = 0xffu;   // max level in the tile
constexpr uint32_t kHdrCountShift = 8;      // bits 8..16: valid slots (0..256)
constexpr uint32_t kHdrCountMask = 0x1ffu;
constexpr uint32_t kHdrExt = 1u << 17;      // some node has an external parent
constexpr uint32_t kHdrWaveLocal = 1u << 18; // every in-tile parent sits in its child's 64-slot group: no workgroup barrier
constexpr uint32_t kHdrAllDynamic = 1u << 19; // every valid slot of the tile carries a Dynamic body (set by bge_world_upload_bodies):
                                              // the tick kernel then loads the velocities without waiting for the flag words
constexpr uint32_t kHdrFrozen = 1u << 20; // some root of the tile keeps its stored world matrix while it is clean (WorldView::frozen has the slots)
constexpr uint32_t kGroup = 64;             // slots per wave64

struct Flattened {
    uint64_t n_entities = 0;
    uint64_t n_transforms = 0;
    uint64_t n_slots = 0;
    uint32_t n_tiles_ticked = 0; // tiles [0, n_tiles_ticked) are launched; the rest is limbo storage
    uint32_t n_tiles_total = 0;
    uint64_t n_limbo = 0;
    bool identity = false;       // slot == entity index for every entity (flat scenes): copies need no gather / scatter
    uint32_t max_depth = 0; // deepest node (global depth, root = 0)
    std::vector<uint32_t> slot_of_entity;   // [n_entities] or kNone
    std::vector<uint32_t> entity_of_slot;   // [n_slots] or kNone
    std::vector<uint32_t> parent_field;     // [n_slots] in-tile index, global slot (kExtParent) or kNone
    std::vector<uint32_t> flags;            // [n_slots] structural bits only (valid/parent/level/limbo)
    std::vector<uint32_t> tile_hdr;         // [n_tiles_total]
    std::vector<uint32_t> pass_tile_begin;  // [n_passes + 1] tile ranges per pass
    std::vector<uint32_t> root_slots;       // slots of roots, in entity order
    std::vector<uint32_t> pass_of_entity;   // [n_entities] (kNone for no-transform / limbo)
};

struct FlattenOptions {
    // Subtrees of <= 64 nodes and at most this many levels below their root are packed into wave-local groups
    // (no workgroup barrier, but the level loop runs with only that level's lanes of the group active).
    // Measured on MI355X: with 25 % of the lanes per level (depth-4 chains, 64-node subtrees) the level-major
    // block layout is faster (kernel is close to VALU-bound), so the default keeps only singletons wave-local.
    uint32_t wave_local_max_height = 0;
};
FlattenOptions flatten_options_from_env(); // BGE_WAVE_LOCAL_HEIGHT overrides the default (experiments)

// parent[i] == kNone or >= n means "no parent".  has_transform may be null (all true).
void flatten_topology(uint64_t n, const uint32_t* parent, const uint8_t* has_transform, Flattened& out,
                      const FlattenOptions& opt = flatten_options_from_env());

// Greedy whole-subtree partition (largest first).  rank_of_entity[n]; nodes_per_rank[nranks].
void partition_subtrees(uint64_t n, const uint32_t* parent, const uint8_t* has_transform, uint32_t nranks,
                        uint32_t* rank_of_entity, uint64_t* nodes_per_rank);

} // namespace bge
