// dmt_json.hpp -- small JSON reader for the scene loader (objects keep their members SORTED BY KEY, which is the
// iteration order of the reference's nlohmann::json objects: src/core/private/core-parser.cpp walks "world" with
// .items(), so instance and light order follows key order, not file order).
#pragma once

#include <cstdlib>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace dmt_host {
namespace json {

struct Value {
  enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
  bool boolean = false;
  double number = 0.0;
  bool integer = false;  // the token had no fraction / exponent (nlohmann's is_number_integer)
  std::string string;
  std::vector<Value> array;
  std::map<std::string, Value> object;

  bool isNumber() const { return kind == Number; }
  bool isInteger() const { return kind == Number && integer; }
  bool isString() const { return kind == String; }
  bool isArray() const { return kind == Array; }
  bool isObject() const { return kind == Object; }
  bool isNull() const { return kind == Null; }
  bool contains(std::string const& k) const { return kind == Object && object.count(k) != 0; }
  Value const& at(std::string const& k) const { return object.at(k); }
  size_t size() const { return kind == Object ? object.size() : (kind == Array ? array.size() : 0); }
};

class Reader {
 public:
  explicit Reader(std::string const& text) : s_(text) {}
  bool parse(Value& out, std::string& err) {
    skip();
    if (!value(out, err)) return false;
    skip();
    if (p_ != s_.size()) return fail(err, "trailing characters");
    return true;
  }

 private:
  std::string const& s_;
  size_t p_ = 0;

  bool fail(std::string& err, char const* what) {
    err = std::string("JSON: ") + what + " at offset " + std::to_string(p_);
    return false;
  }
  void skip() {
    while (p_ < s_.size() && (s_[p_] == ' ' || s_[p_] == '\t' || s_[p_] == '\n' || s_[p_] == '\r')) ++p_;
  }
  bool literal(char const* lit) {
    size_t n = 0;
    while (lit[n]) ++n;
    if (s_.compare(p_, n, lit) != 0) return false;
    p_ += n;
    return true;
  }
  bool string(std::string& out, std::string& err) {
    if (s_[p_] != '"') return fail(err, "expected string");
    ++p_;
    out.clear();
    while (p_ < s_.size() && s_[p_] != '"') {
      char c = s_[p_++];
      if (c == '\\') {
        if (p_ >= s_.size()) return fail(err, "bad escape");
        char const e = s_[p_++];
        switch (e) {
          case '"': out += '"'; break;
          case '\\': out += '\\'; break;
          case '/': out += '/'; break;
          case 'b': out += '\b'; break;
          case 'f': out += '\f'; break;
          case 'n': out += '\n'; break;
          case 'r': out += '\r'; break;
          case 't': out += '\t'; break;
          case 'u': {
            if (p_ + 4 > s_.size()) return fail(err, "bad \\u escape");
            unsigned const cp = unsigned(std::strtoul(s_.substr(p_, 4).c_str(), nullptr, 16));
            p_ += 4;
            if (cp < 0x80) out += char(cp);
            else if (cp < 0x800) out += char(0xC0 | (cp >> 6)), out += char(0x80 | (cp & 0x3F));
            else out += char(0xE0 | (cp >> 12)), out += char(0x80 | ((cp >> 6) & 0x3F)), out += char(0x80 | (cp & 0x3F));
            break;
          }
          default: return fail(err, "bad escape");
        }
      } else {
        out += c;
      }
    }
    if (p_ >= s_.size()) return fail(err, "unterminated string");
    ++p_;
    return true;
  }
  bool value(Value& v, std::string& err) {
    if (p_ >= s_.size()) return fail(err, "unexpected end");
    char const c = s_[p_];
    if (c == '{') {
      v.kind = Value::Object;
      ++p_;
      skip();
      if (p_ < s_.size() && s_[p_] == '}') return ++p_, true;
      for (;;) {
        skip();
        std::string key;
        if (p_ >= s_.size() || !string(key, err)) return err.empty() ? fail(err, "expected key") : false;
        skip();
        if (p_ >= s_.size() || s_[p_] != ':') return fail(err, "expected ':'");
        ++p_;
        skip();
        Value child;
        if (!value(child, err)) return false;
        v.object[key] = std::move(child);  // a repeated key keeps the last value, as nlohmann does
        skip();
        if (p_ < s_.size() && s_[p_] == ',') { ++p_; continue; }
        if (p_ < s_.size() && s_[p_] == '}') return ++p_, true;
        return fail(err, "expected ',' or '}'");
      }
    }
    if (c == '[') {
      v.kind = Value::Array;
      ++p_;
      skip();
      if (p_ < s_.size() && s_[p_] == ']') return ++p_, true;
      for (;;) {
        skip();
        Value child;
        if (!value(child, err)) return false;
        v.array.push_back(std::move(child));
        skip();
        if (p_ < s_.size() && s_[p_] == ',') { ++p_; continue; }
        if (p_ < s_.size() && s_[p_] == ']') return ++p_, true;
        return fail(err, "expected ',' or ']'");
      }
    }
    if (c == '"') {
      v.kind = Value::String;
      return string(v.string, err);
    }
    if (literal("true")) return v.kind = Value::Bool, v.boolean = true, true;
    if (literal("false")) return v.kind = Value::Bool, v.boolean = false, true;
    if (literal("null")) return v.kind = Value::Null, true;
    if (c == '-' || (c >= '0' && c <= '9')) {
      size_t const start = p_;
      bool integer = true;
      if (s_[p_] == '-') ++p_;
      while (p_ < s_.size() && ((s_[p_] >= '0' && s_[p_] <= '9') || s_[p_] == '.' || s_[p_] == 'e' || s_[p_] == 'E' ||
                                s_[p_] == '+' || s_[p_] == '-')) {
        if (s_[p_] == '.' || s_[p_] == 'e' || s_[p_] == 'E') integer = false;
        ++p_;
      }
      v.kind = Value::Number;
      v.integer = integer;
      v.number = std::strtod(s_.substr(start, p_ - start).c_str(), nullptr);
      return true;
    }
    return fail(err, "unexpected character");
  }
};

}  // namespace json
}  // namespace dmt_host
