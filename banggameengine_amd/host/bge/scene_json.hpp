// bge/scene_json.hpp — ingest of the reference's on-disk scene format (assets/scenes/*.json) into a scene store.
//
// Follows the ECS-relevant part of LoadSceneFromJson (src/scene/SceneLoader.cpp:652-745):
//   entities[]            ProcessEntityJson (:585-648): CreateEntity, AddTransform, then optional components
//   transform             ApplyTransformFromJson (:435-504): "position", "rotationEuler" (radians) or
//                         "rotationEulerDeg" (bx::toRad = deg * kPi / 180, wins when both are present), "scale";
//                         partial arrays keep the defaults of the untouched elements; always MarkDirty
//   collider              ApplyColliderFromJson (:208-232): "shape" box|capsule (case-insensitive, unknown -> box),
//                         box "size" [hx,hy,hz]; capsule "radius", "height" (stored as size.x, size.y = height/2)
//   rigidBody             ApplyRigidBodyFromJson (:234-271): "type" static|dynamic|kinematic (default Static),
//                         mass (Dynamic: default 1, else 0), friction, restitution, layer, mask (numbers or strings
//                         parsed with base auto-detection)
//   trigger               ApplyTriggerFromJson (:273-301): shape / size as for colliders, "layer" (default: the component's
//                         own value, or 1 << 2 when that is 0), "mask", "oneShot", "active" (default TRUE whatever the
//                         component held); only on scene types that have AddTriggerVolume
//   children[] / parent   nested children are parented to the enclosing entity unless they carry a "parent" key;
//                         "parent" strings (an entity's "name" or "id") are resolved after all entities exist (:727-738)
// Everything else in the file (resources, meshRenderer, …) is outside the world tick and ignored.
// Templated on the scene type like gpu_systems.hpp: works on bge::Scene and on the reference's Scene.
// nlohmann_json (the reference's parser) is not available here, so a small strict JSON reader is included.
#pragma once

#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <string>
#include <type_traits>
#include <unordered_map>
#include <utility>
#include <vector>

namespace bge {
namespace json {

struct Value {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0.0;
    bool is_integer = false; // written without fraction / exponent
    std::string str;
    std::vector<Value> arr;
    std::vector<std::pair<std::string, Value>> obj;

    const Value* find(const char* key) const
    {
        if (kind != Object) return nullptr;
        for (const auto& kv : obj) {
            if (kv.first == key) return &kv.second;
        }
        return nullptr;
    }
    bool is_number() const { return kind == Number; }
};

class Parser {
public:
    explicit Parser(const std::string& text) : s_(text) {}
    bool parse(Value& out, std::string* err)
    {
        skip();
        if (!value(out)) return fail(err);
        skip();
        if (p_ != s_.size()) {
            msg_ = "trailing characters";
            return fail(err);
        }
        return true;
    }

private:
    bool fail(std::string* err)
    {
        if (err) *err = "JSON error at offset " + std::to_string(p_) + ": " + msg_;
        return false;
    }
    void skip()
    {
        while (p_ < s_.size() && (s_[p_] == ' ' || s_[p_] == '\n' || s_[p_] == '\r' || s_[p_] == '\t')) ++p_;
    }
    bool literal(const char* lit)
    {
        size_t n = 0;
        while (lit[n]) ++n;
        if (s_.compare(p_, n, lit) != 0) {
            msg_ = "unexpected token";
            return false;
        }
        p_ += n;
        return true;
    }
    bool value(Value& v)
    {
        if (p_ >= s_.size()) {
            msg_ = "unexpected end";
            return false;
        }
        const char c = s_[p_];
        if (c == '{' || c == '[') {
            // bounded nesting: the parser is recursive, and a file of 100 000 '[' must be an error, not a stack overflow
            // (found by tests/cpp/sanitize_host.cpp); the reference's scenes nest 4 levels deep
            if (depth_ >= kMaxDepth) {
                msg_ = "nesting deeper than " + std::to_string(kMaxDepth);
                return false;
            }
            ++depth_;
            const bool ok = c == '{' ? object(v) : array(v);
            --depth_;
            return ok;
        }
        if (c == '"') {
            v.kind = Value::String;
            return string(v.str);
        }
        if (c == 't') {
            v.kind = Value::Bool;
            v.b = true;
            return literal("true");
        }
        if (c == 'f') {
            v.kind = Value::Bool;
            v.b = false;
            return literal("false");
        }
        if (c == 'n') {
            v.kind = Value::Null;
            return literal("null");
        }
        return number(v);
    }
    bool number(Value& v)
    {
        const size_t start = p_;
        bool integer = true;
        if (p_ < s_.size() && s_[p_] == '-') ++p_;
        while (p_ < s_.size() && std::isdigit(static_cast<unsigned char>(s_[p_]))) ++p_;
        if (p_ < s_.size() && s_[p_] == '.') {
            integer = false;
            ++p_;
            while (p_ < s_.size() && std::isdigit(static_cast<unsigned char>(s_[p_]))) ++p_;
        }
        if (p_ < s_.size() && (s_[p_] == 'e' || s_[p_] == 'E')) {
            integer = false;
            ++p_;
            if (p_ < s_.size() && (s_[p_] == '+' || s_[p_] == '-')) ++p_;
            while (p_ < s_.size() && std::isdigit(static_cast<unsigned char>(s_[p_]))) ++p_;
        }
        if (p_ == start) {
            msg_ = "invalid value";
            return false;
        }
        v.kind = Value::Number;
        v.is_integer = integer;
        v.num = std::strtod(s_.substr(start, p_ - start).c_str(), nullptr);
        return true;
    }
    bool string(std::string& out)
    {
        ++p_; // opening quote
        out.clear();
        while (p_ < s_.size() && s_[p_] != '"') {
            char c = s_[p_++];
            if (c == '\\') {
                if (p_ >= s_.size()) break;
                const char e = s_[p_++];
                switch (e) {
                case 'n': c = '\n'; break;
                case 't': c = '\t'; break;
                case 'r': c = '\r'; break;
                case 'b': c = '\b'; break;
                case 'f': c = '\f'; break;
                case 'u': { // keep BMP code points as UTF-8
                    if (p_ + 4 > s_.size()) {
                        msg_ = "bad \\u escape";
                        return false;
                    }
                    const unsigned cp = static_cast<unsigned>(std::strtoul(s_.substr(p_, 4).c_str(), nullptr, 16));
                    p_ += 4;
                    if (cp < 0x80) {
                        out.push_back(static_cast<char>(cp));
                    } else if (cp < 0x800) {
                        out.push_back(static_cast<char>(0xC0 | (cp >> 6)));
                        out.push_back(static_cast<char>(0x80 | (cp & 0x3F)));
                    } else {
                        out.push_back(static_cast<char>(0xE0 | (cp >> 12)));
                        out.push_back(static_cast<char>(0x80 | ((cp >> 6) & 0x3F)));
                        out.push_back(static_cast<char>(0x80 | (cp & 0x3F)));
                    }
                    continue;
                }
                default: c = e; break; // \" \\ \/
                }
            }
            out.push_back(c);
        }
        if (p_ >= s_.size()) {
            msg_ = "unterminated string";
            return false;
        }
        ++p_; // closing quote
        return true;
    }
    bool array(Value& v)
    {
        v.kind = Value::Array;
        ++p_;
        skip();
        if (p_ < s_.size() && s_[p_] == ']') {
            ++p_;
            return true;
        }
        for (;;) {
            v.arr.emplace_back();
            skip();
            if (!value(v.arr.back())) return false;
            skip();
            if (p_ < s_.size() && s_[p_] == ',') {
                ++p_;
                continue;
            }
            if (p_ < s_.size() && s_[p_] == ']') {
                ++p_;
                return true;
            }
            msg_ = "expected ',' or ']'";
            return false;
        }
    }
    bool object(Value& v)
    {
        v.kind = Value::Object;
        ++p_;
        skip();
        if (p_ < s_.size() && s_[p_] == '}') {
            ++p_;
            return true;
        }
        for (;;) {
            skip();
            if (p_ >= s_.size() || s_[p_] != '"') {
                msg_ = "expected a key";
                return false;
            }
            std::string key;
            if (!string(key)) return false;
            skip();
            if (p_ >= s_.size() || s_[p_] != ':') {
                msg_ = "expected ':'";
                return false;
            }
            ++p_;
            skip();
            v.obj.emplace_back(std::move(key), Value{});
            if (!value(v.obj.back().second)) return false;
            skip();
            if (p_ < s_.size() && s_[p_] == ',') {
                ++p_;
                continue;
            }
            if (p_ < s_.size() && s_[p_] == '}') {
                ++p_;
                return true;
            }
            msg_ = "expected ',' or '}'";
            return false;
        }
    }

    const std::string& s_;
    size_t p_ = 0;
    static constexpr int kMaxDepth = 128;
    int depth_ = 0;
    std::string msg_;
};

} // namespace json

namespace detail {

inline std::string lower(std::string s)
{
    for (char& c : s) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
    return s;
}
inline float read_float(const json::Value& parent, const char* key, float fallback)
{
    const json::Value* v = parent.find(key);
    return v && v->is_number() ? static_cast<float>(v->num) : fallback;
}
inline uint32_t read_uint(const json::Value& parent, const char* key, uint32_t fallback)
{
    const json::Value* v = parent.find(key);
    if (!v) return fallback;
    if (v->is_number() && v->is_integer) return v->num < 0 ? 0u : static_cast<uint32_t>(static_cast<int64_t>(v->num));
    if (v->kind == json::Value::String) {
        char* end = nullptr;
        const unsigned long x = std::strtoul(v->str.c_str(), &end, 0); // base auto-detect, as std::stoul(text, &idx, 0)
        if (end != v->str.c_str()) return static_cast<uint32_t>(x);
    }
    return fallback;
}
inline bool read_bool(const json::Value& parent, const char* key, bool fallback)
{
    const json::Value* v = parent.find(key);
    return v && v->kind == json::Value::Bool ? v->b : fallback; // (nlohmann's value() throws on a non-bool; here it is ignored)
}
template <class S, class = void> struct can_add_trigger : std::false_type {};
template <class S> struct can_add_trigger<S, std::void_t<decltype(std::declval<S&>().AddTriggerVolume(0u))>> : std::true_type {};
inline std::string read_string(const json::Value& parent, const char* key, const std::string& fallback)
{
    const json::Value* v = parent.find(key);
    return v && v->kind == json::Value::String ? v->str : fallback;
}
// readVec3 of ApplyTransformFromJson: returns whether any element was taken
template <class V3> bool read_vec3(const json::Value& parent, const char* key, V3& inout)
{
    const json::Value* v = parent.find(key);
    if (!v || v->kind != json::Value::Array) return false;
    bool modified = false;
    float* dst[3] = {&inout.x, &inout.y, &inout.z};
    for (size_t i = 0; i < 3 && i < v->arr.size(); ++i) {
        if (v->arr[i].is_number()) {
            *dst[i] = static_cast<float>(v->arr[i].num);
            modified = true;
        }
    }
    return modified;
}
inline float to_rad(float deg) { return deg * 3.1415926535897932384626433832795f / 180.0f; } // bx::toRad

} // namespace detail

// Builds entities of `scene` from the text of a scene file.  `lookup` (optional) receives name/id -> EntityId.
template <class SceneT>
bool LoadSceneFromJsonText(const std::string& text, SceneT& scene, std::string* err = nullptr,
                           std::unordered_map<std::string, uint32_t>* lookup = nullptr)
{
    json::Value root;
    json::Parser parser(text);
    if (!parser.parse(root, err)) return false;
    if (root.kind != json::Value::Object) {
        if (err) *err = "scene root must be an object";
        return false;
    }
    const json::Value* entities = root.find("entities");
    if (!entities) return true;
    if (entities->kind != json::Value::Array) {
        if (err) *err = "'entities' must be an array";
        return false;
    }
    std::unordered_map<std::string, uint32_t> keys;
    std::vector<std::pair<uint32_t, std::string>> pending_parents;
    unsigned auto_names = 0;

    struct Walker {
        SceneT& scene;
        std::unordered_map<std::string, uint32_t>& keys;
        std::vector<std::pair<uint32_t, std::string>>& pending;
        unsigned& auto_names;

        void entity(const json::Value& e, uint32_t forced_parent)
        {
            const uint32_t id = scene.CreateEntity();
            const std::string name = detail::read_string(e, "name", "");
            const std::string explicit_id = detail::read_string(e, "id", "");
            if (!name.empty()) keys[name] = id;             // duplicates overwrite, as RegisterEntityKey does
            if (!explicit_id.empty()) keys[explicit_id] = id;
            if (name.empty() && explicit_id.empty()) keys["__entity_" + std::to_string(auto_names++)] = id;

            if (auto* t = scene.AddTransform(id)) {
                static const json::Value empty_object = [] { json::Value v; v.kind = json::Value::Object; return v; }();
                const json::Value* tj = e.find("transform");
                if (!tj || tj->kind != json::Value::Object) tj = &empty_object;
                detail::read_vec3(*tj, "position", t->position);
                auto rot = t->rotationEuler;
                bool has_rot = detail::read_vec3(*tj, "rotationEuler", rot);
                auto deg = t->rotationEuler;
                if (detail::read_vec3(*tj, "rotationEulerDeg", deg)) {
                    rot.x = detail::to_rad(deg.x);
                    rot.y = detail::to_rad(deg.y);
                    rot.z = detail::to_rad(deg.z);
                    has_rot = true;
                }
                if (has_rot) t->rotationEuler = rot;
                detail::read_vec3(*tj, "scale", t->scale);
                t->MarkDirty();
            }
            if (const json::Value* cj = e.find("collider"); cj && cj->kind == json::Value::Object) {
                if (auto* c = scene.AddCollider(id)) {
                    const std::string shape = detail::lower(detail::read_string(*cj, "shape", "box"));
                    using ShapeT = decltype(c->shape);
                    c->shape = static_cast<ShapeT>(shape == "capsule" ? 1 : 0); // unknown shapes fall back to box
                    if (shape != "capsule") {
                        detail::read_vec3(*cj, "size", c->size); // missing or non-array "size": defaults stay
                    } else {
                        const float radius = detail::read_float(*cj, "radius", c->size.x);
                        const float height = detail::read_float(*cj, "height", c->size.y * 2.0f);
                        c->size.x = radius;
                        c->size.y = height * 0.5f;
                    }
                    c->dirty = true;
                }
            }
            if (const json::Value* rj = e.find("rigidBody"); rj && rj->kind == json::Value::Object) {
                if (auto* b = scene.AddRigidBody(id)) {
                    const std::string type = detail::lower(detail::read_string(*rj, "type", "Static"));
                    using TypeT = decltype(b->type);
                    const int t = type == "dynamic" ? 1 : (type == "kinematic" ? 2 : 0);
                    b->type = static_cast<TypeT>(t);
                    b->mass = t == 1 ? detail::read_float(*rj, "mass", 1.0f) : 0.0f;
                    b->friction = detail::read_float(*rj, "friction", b->friction);
                    b->restitution = detail::read_float(*rj, "restitution", b->restitution);
                    b->layer = detail::read_uint(*rj, "layer", b->layer);
                    b->mask = detail::read_uint(*rj, "mask", b->mask);
                    b->dirty = true;
                }
            }
            if constexpr (detail::can_add_trigger<SceneT>::value) {
                if (const json::Value* tj = e.find("trigger"); tj && tj->kind == json::Value::Object) {
                    if (auto* tv = scene.AddTriggerVolume(id)) {
                        const std::string shape = detail::lower(detail::read_string(*tj, "shape", "box"));
                        using ShapeT = decltype(tv->shape);
                        tv->shape = static_cast<ShapeT>(shape == "capsule" ? 1 : 0);
                        if (shape != "capsule") {
                            detail::read_vec3(*tj, "size", tv->size);
                        } else {
                            const float radius = detail::read_float(*tj, "radius", tv->size.x);
                            const float height = detail::read_float(*tj, "height", tv->size.y * 2.0f);
                            tv->size.x = radius;
                            tv->size.y = height * 0.5f;
                        }
                        tv->layer = detail::read_uint(*tj, "layer", tv->layer ? tv->layer : (1u << 2)); // kDefaultTriggerLayer
                        tv->mask = detail::read_uint(*tj, "mask", tv->mask);
                        tv->oneShot = detail::read_bool(*tj, "oneShot", tv->oneShot);
                        tv->active = detail::read_bool(*tj, "active", true);
                        tv->dirty = true;
                    }
                }
            }
            if (const json::Value* pj = e.find("parent"); pj && pj->kind == json::Value::String) {
                pending.emplace_back(id, pj->str);
            } else if (forced_parent != 0) {
                scene.SetParent(id, forced_parent);
            }
            if (const json::Value* kids = e.find("children"); kids && kids->kind == json::Value::Array) {
                for (const json::Value& k : kids->arr) {
                    if (k.kind == json::Value::Object) entity(k, id);
                }
            }
        }
    } walker{scene, keys, pending_parents, auto_names};

    for (const json::Value& e : entities->arr) {
        if (e.kind == json::Value::Object) walker.entity(e, 0);
    }
    for (const auto& pr : pending_parents) {
        auto it = keys.find(pr.second);
        if (it != keys.end()) scene.SetParent(pr.first, it->second); // unknown parents are reported and skipped upstream
    }
    if (lookup) *lookup = std::move(keys);
    return true;
}

} // namespace bge
