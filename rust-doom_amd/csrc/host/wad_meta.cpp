// wad::meta: the metadata TOML file (`--metadata`, e.g. rust-doom's assets/meta/doom.toml) ->
// WadMetadata.  Reference: wad/src/meta.rs:15-257.  The TOML reader below covers the subset those
// files use: comments, [tables], [[arrays of tables]] with nested sub-tables, dotted headers, bare
// and quoted keys, basic/literal strings, integers, floats, booleans, (multi-line) arrays and inline
// tables.  Unknown keys are ignored, as serde does for structs without deny_unknown_fields.
#include <cstring>
#include <fstream>
#include <mutex>
#include <sstream>

#include "wad.hpp"

namespace rdoom::wad {
namespace {

struct TomlValue {
  enum Type { None, Str, Int, Float, Bool, Array, Table } type = None;
  std::string s;
  int64_t i = 0;
  double f = 0;
  bool b = false;
  std::vector<TomlValue> arr;
  std::vector<std::pair<std::string, TomlValue>> tbl;

  TomlValue *find(const std::string &k) {
    for (auto &kv : tbl)
      if (kv.first == k) return &kv.second;
    return nullptr;
  }
  const TomlValue *find(const std::string &k) const { return const_cast<TomlValue *>(this)->find(k); }
  TomlValue &get_or_insert(const std::string &k, Type t) {
    if (TomlValue *v = find(k)) return *v;
    tbl.emplace_back(k, TomlValue{});
    tbl.back().second.type = t;
    return tbl.back().second;
  }
};

class TomlParser {
 public:
  explicit TomlParser(const std::string &text) : t_(text) {}

  TomlValue parse() {
    TomlValue root;
    root.type = TomlValue::Table;
    TomlValue *cur = &root;
    for (;;) {
      skip_ws_nl();
      if (eof()) break;
      if (peek() == '[') {
        const bool aot = t_.compare(p_, 2, "[[") == 0;
        p_ += aot ? 2 : 1;
        std::vector<std::string> path = key_path();
        skip_ws();
        expect(']');
        if (aot) expect(']');
        cur = &root;
        for (size_t k = 0; k < path.size(); k++) {
          const bool last = k + 1 == path.size();
          if (last && aot) {
            TomlValue &arr = cur->get_or_insert(path[k], TomlValue::Array);
            if (arr.type != TomlValue::Array) fail("key redefined as array of tables");
            arr.arr.emplace_back();
            arr.arr.back().type = TomlValue::Table;
            cur = &arr.arr.back();
          } else {
            TomlValue &next = cur->get_or_insert(path[k], TomlValue::Table);
            if (next.type == TomlValue::Array) {
              if (next.arr.empty() || next.arr.back().type != TomlValue::Table) fail("bad table path");
              cur = &next.arr.back();
            } else if (next.type == TomlValue::Table) {
              cur = &next;
            } else {
              fail("table path crosses a value");
            }
          }
        }
        end_of_line();
      } else {
        key_value(*cur);
        end_of_line();
      }
    }
    return root;
  }

 private:
  const std::string &t_;
  size_t p_ = 0;

  bool eof() const { return p_ >= t_.size(); }
  char peek() const { return eof() ? '\0' : t_[p_]; }
  [[noreturn]] void fail(const std::string &why) const {
    size_t line = 1;
    for (size_t i = 0; i < p_ && i < t_.size(); i++)
      if (t_[i] == '\n') line++;
    throw WadError(RDOOM_CORRUPT_META, "metadata parse error at line " + std::to_string(line) + ": " + why);
  }
  void expect(char c) {
    if (peek() != c) fail(std::string("expected '") + c + "'");
    p_++;
  }
  void skip_ws() {
    while (!eof() && (t_[p_] == ' ' || t_[p_] == '\t')) p_++;
  }
  void skip_comment() {
    if (peek() == '#')
      while (!eof() && t_[p_] != '\n') p_++;
  }
  void skip_ws_nl() {
    for (;;) {
      skip_ws();
      skip_comment();
      if (!eof() && (t_[p_] == '\n' || t_[p_] == '\r'))
        p_++;
      else
        break;
    }
  }
  void end_of_line() {
    skip_ws();
    skip_comment();
    if (eof()) return;
    if (t_[p_] == '\r') p_++;
    if (peek() != '\n') fail("expected end of line");
    p_++;
  }
  std::string key() {
    skip_ws();
    if (peek() == '"') return basic_string();
    if (peek() == '\'') return literal_string();
    const size_t s = p_;
    while (!eof() && (isalnum((unsigned char)t_[p_]) || t_[p_] == '_' || t_[p_] == '-')) p_++;
    if (p_ == s) fail("expected a key");
    return t_.substr(s, p_ - s);
  }
  std::vector<std::string> key_path() {
    std::vector<std::string> path;
    for (;;) {
      path.push_back(key());
      skip_ws();
      if (peek() == '.') {
        p_++;
        continue;
      }
      break;
    }
    return path;
  }
  void key_value(TomlValue &table) {
    std::vector<std::string> path = key_path();
    skip_ws();
    expect('=');
    TomlValue *cur = &table;
    for (size_t k = 0; k + 1 < path.size(); k++) {
      cur = &cur->get_or_insert(path[k], TomlValue::Table);
      if (cur->type != TomlValue::Table) fail("dotted key crosses a value");
    }
    TomlValue v = value();
    if (cur->find(path.back())) fail("duplicate key '" + path.back() + "'");
    cur->tbl.emplace_back(path.back(), std::move(v));
  }
  std::string basic_string() {
    expect('"');
    if (t_.compare(p_, 2, "\"\"") == 0) fail("multi-line strings are not supported");
    std::string out;
    while (!eof() && t_[p_] != '"') {
      char c = t_[p_++];
      if (c == '\n') fail("newline in string");
      if (c == '\\') {
        if (eof()) fail("bad escape");
        const char e = t_[p_++];
        switch (e) {
          case 'n': out.push_back('\n'); break;
          case 't': out.push_back('\t'); break;
          case 'r': out.push_back('\r'); break;
          case '\\': out.push_back('\\'); break;
          case '"': out.push_back('"'); break;
          default: fail("unsupported escape");
        }
      } else {
        out.push_back(c);
      }
    }
    expect('"');
    return out;
  }
  std::string literal_string() {
    expect('\'');
    const size_t s = p_;
    while (!eof() && t_[p_] != '\'' && t_[p_] != '\n') p_++;
    std::string out = t_.substr(s, p_ - s);
    expect('\'');
    return out;
  }
  TomlValue value() {
    skip_ws();
    TomlValue v;
    const char c = peek();
    if (c == '"') {
      v.type = TomlValue::Str;
      v.s = basic_string();
    } else if (c == '\'') {
      v.type = TomlValue::Str;
      v.s = literal_string();
    } else if (c == '[') {
      p_++;
      v.type = TomlValue::Array;
      for (;;) {
        skip_ws_nl();
        if (peek() == ']') {
          p_++;
          break;
        }
        v.arr.push_back(value());
        skip_ws_nl();
        if (peek() == ',') {
          p_++;
          continue;
        }
        skip_ws_nl();
        expect(']');
        break;
      }
    } else if (c == '{') {
      p_++;
      v.type = TomlValue::Table;
      for (;;) {
        skip_ws_nl();
        if (peek() == '}') {
          p_++;
          break;
        }
        key_value(v);
        skip_ws_nl();
        if (peek() == ',') {
          p_++;
          continue;
        }
        expect('}');
        break;
      }
    } else if (t_.compare(p_, 4, "true") == 0) {
      p_ += 4;
      v.type = TomlValue::Bool;
      v.b = true;
    } else if (t_.compare(p_, 5, "false") == 0) {
      p_ += 5;
      v.type = TomlValue::Bool;
      v.b = false;
    } else {
      const size_t s = p_;
      bool is_float = false;
      while (!eof() && (isalnum((unsigned char)t_[p_]) || t_[p_] == '+' || t_[p_] == '-' || t_[p_] == '.' ||
                        t_[p_] == '_')) {
        if (t_[p_] == '.' || t_[p_] == 'e' || t_[p_] == 'E') is_float = true;
        p_++;
      }
      std::string num;
      for (size_t k = s; k < p_; k++)
        if (t_[k] != '_') num.push_back(t_[k]);
      if (num.empty()) fail("expected a value");
      if (num == "inf" || num == "+inf" || num == "-inf" || num == "nan") is_float = true;
      try {
        size_t used = 0;
        if (is_float) {
          v.type = TomlValue::Float;
          v.f = std::stod(num, &used);
        } else {
          v.type = TomlValue::Int;
          v.i = std::stoll(num, &used, 10);
        }
        if (used != num.size()) fail("bad number '" + num + "'");
      } catch (const WadError &) {
        throw;
      } catch (...) {
        fail("bad number '" + num + "'");
      }
    }
    return v;
  }
};

[[noreturn]] void meta_fail(const std::string &why) { throw WadError(RDOOM_CORRUPT_META, "metadata: " + why); }

const TomlValue &need(const TomlValue &t, const char *k, TomlValue::Type ty) {
  const TomlValue *v = t.find(k);
  if (!v) meta_fail(std::string("missing field `") + k + "`");
  if (v->type != ty && !(ty == TomlValue::Float && v->type == TomlValue::Int))
    meta_fail(std::string("field `") + k + "` has the wrong type");
  return *v;
}
double num(const TomlValue &v) { return v.type == TomlValue::Int ? (double)v.i : v.f; }

WadName name_from(const TomlValue &v) {
  try {
    return WadName::from_str(v.s);
  } catch (const WadError &e) {
    meta_fail(std::string("bad name '") + v.s + "': " + e.what());
  }
}

std::vector<std::vector<WadName>> names2(const TomlValue &v) {
  std::vector<std::vector<WadName>> out;
  for (const TomlValue &a : v.arr) {
    if (a.type != TomlValue::Array) meta_fail("animations must be arrays of arrays of strings");
    out.emplace_back();
    for (const TomlValue &n : a.arr) {
      if (n.type != TomlValue::Str) meta_fail("animation frame must be a string");
      out.back().push_back(name_from(n));
    }
  }
  return out;
}

HeightDef height_def(const TomlValue &t) {
  static const std::pair<const char *, HeightRef> refs[] = {
      {"LowestFloor", HeightRef::LowestFloor},     {"NextFloor", HeightRef::NextFloor},
      {"HighestFloor", HeightRef::HighestFloor},   {"LowestCeiling", HeightRef::LowestCeiling},
      {"HighestCeiling", HeightRef::HighestCeiling}, {"Floor", HeightRef::Floor},
      {"Ceiling", HeightRef::Ceiling}};
  if (t.type != TomlValue::Table) meta_fail("height definition must be a table");
  const std::string &to = need(t, "to", TomlValue::Str).s;
  HeightDef d{};
  bool found = false;
  for (auto &r : refs)
    if (to == r.first) {
      d.to = r.second;
      found = true;
    }
  if (!found) meta_fail("unknown height reference '" + to + "'");
  if (const TomlValue *off = t.find("off")) {
    if (off->type != TomlValue::Int) meta_fail("`off` must be an integer");
    d.offset = (int16_t)off->i;
  }
  return d;
}

HeightEffectDef height_effect(const TomlValue &t) {
  if (t.type != TomlValue::Table) meta_fail("height effect must be a table");
  HeightEffectDef e;
  const TomlValue *first = t.find("first");
  if (!first) meta_fail("missing field `first`");
  e.first = height_def(*first);
  if (const TomlValue *second = t.find("second")) e.second = height_def(*second);
  return e;
}

}  // namespace

static std::mutex &regex_mutex() {
  static std::mutex m;
  return m;
}

WadMetadata WadMetadata::from_file(const std::string &path) {
  std::ifstream f(path, std::ios::binary);
  if (!f) throw WadError(RDOOM_IO, "cannot read metadata file '" + path + "'");
  std::stringstream ss;
  ss << f.rdbuf();
  return from_text(ss.str());
}

WadMetadata WadMetadata::from_text(const std::string &text) {
  const TomlValue root = TomlParser(text).parse();
  WadMetadata m;
  for (const TomlValue &s : need(root, "sky", TomlValue::Array).arr) {
    SkyMetadata sky;
    sky.texture_name = name_from(need(s, "texture_name", TomlValue::Str));
    sky.pattern_text = need(s, "level_pattern", TomlValue::Str).s;
    try {
      // libstdc++'s regex compiler and matcher go through std::ctype<char>::narrow, which fills a cache inside the GLOBAL
      // locale's facet without synchronisation: two host threads opening IWADs at the same time race on it (found by
      // tests/test_host_threads.py under ThreadSanitizer).  Every std::regex operation of the library holds this lock.
      std::lock_guard<std::mutex> lock(regex_mutex());
      sky.level_pattern = std::regex(sky.pattern_text, std::regex::ECMAScript);
    } catch (const std::regex_error &e) {
      meta_fail("bad level_pattern '" + sky.pattern_text + "'");
    }
    sky.tiled_band_size = (float)num(need(s, "tiled_band_size", TomlValue::Float));
    m.sky.push_back(std::move(sky));
  }
  const TomlValue &anim = need(root, "animations", TomlValue::Table);
  m.animated_flats = names2(need(anim, "flats", TomlValue::Array));
  m.animated_walls = names2(need(anim, "walls", TomlValue::Array));
  const TomlValue &things = need(root, "things", TomlValue::Table);
  for (const char *cat : {"decorations", "weapons", "powerups", "artifacts", "ammo", "keys", "monsters"}) {
    for (const TomlValue &t : need(things, cat, TomlValue::Array).arr) {
      ThingMetadata tm;
      tm.thing_type = (uint16_t)need(t, "thing_type", TomlValue::Int).i;
      tm.sprite = name_from(need(t, "sprite", TomlValue::Str));
      tm.sequence = need(t, "sequence", TomlValue::Str).s;
      tm.hanging = need(t, "hanging", TomlValue::Bool).b;
      tm.radius = (uint32_t)need(t, "radius", TomlValue::Int).i;
      m.things.push_back(std::move(tm));
    }
  }
  if (const TomlValue *lds = root.find("linedef")) {
    if (lds->type != TomlValue::Array) meta_fail("`linedef` must be an array of tables");
    for (const TomlValue &l : lds->arr) {
      LinedefMetadata lm;
      lm.special_type = (uint16_t)need(l, "special_type", TomlValue::Int).i;
      lm.trigger = need(l, "trigger", TomlValue::Str).s;
      bool ok = false;
      for (const char *t : {"Any", "Push", "Switch", "WalkOver", "Gun"}) ok |= lm.trigger == t;
      if (!ok) meta_fail("unknown trigger '" + lm.trigger + "'");
      if (const TomlValue *v = l.find("monsters")) lm.monsters = v->b;
      if (const TomlValue *v = l.find("only_once")) lm.only_once = v->b;
      if (const TomlValue *mv = l.find("move")) {
        if (mv->type != TomlValue::Table) meta_fail("`move` must be a table");
        MoveEffectDef me;
        if (const TomlValue *v = mv->find("floor")) me.floor = height_effect(*v);
        if (const TomlValue *v = mv->find("ceiling")) me.ceiling = height_effect(*v);
        if (const TomlValue *v = mv->find("repeat")) me.repeat = v->b;
        if (const TomlValue *v = mv->find("wait")) me.wait = (float)num(*v);
        if (const TomlValue *v = mv->find("speed")) me.speed = (float)num(*v) / 8.0f * 0.7f;  // meta.rs:222-227
        lm.move_effect = me;
      }
      if (const TomlValue *v = l.find("exit")) {
        if (v->type != TomlValue::Str || (v->s != "Normal" && v->s != "Secret")) meta_fail("bad `exit` value");
        lm.exit_effect = v->s;
      }
      m.linedef[lm.special_type] = std::move(lm);  // IndexMap collect: the last entry wins
    }
  }
  return m;
}

const ThingMetadata *WadMetadata::find_thing(uint16_t thing_type) const {
  for (const ThingMetadata &t : things)
    if (t.thing_type == thing_type) return &t;
  return nullptr;
}

const SkyMetadata *WadMetadata::sky_for(const WadName &level_name) const {
  const std::string text((const char *)level_name.b.data(), 8);  // WadName::as_ref keeps the NUL padding
  {
    std::lock_guard<std::mutex> lock(regex_mutex());  // (see from_text)
    for (const SkyMetadata &s : sky)
      if (std::regex_search(text, s.level_pattern)) return &s;
  }
  return sky.empty() ? nullptr : &sky[0];
}

}  // namespace rdoom::wad
