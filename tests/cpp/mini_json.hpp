// mini_json.hpp — the few lines of JSON reading the C++ test drivers need for tests/golden/*.json.  TEST INFRASTRUCTURE.
#pragma once
#include <cstdlib>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

struct J {
  enum T { NUL, BOOL, NUM, STR, ARR, OBJ } t = NUL;
  bool b = false;
  double num = 0;
  std::string s;
  std::vector<J> arr;
  std::vector<std::pair<std::string, J>> obj;
  bool is_null() const { return t == NUL; }
  const J &operator[](const std::string &k) const {
    for (auto &kv : obj) if (kv.first == k) return kv.second;
    static const J none; return none;
  }
  bool has(const std::string &k) const { for (auto &kv : obj) if (kv.first == k) return true; return false; }
  const J &operator[](size_t i) const { return arr.at(i); }
  size_t size() const { return t == ARR ? arr.size() : obj.size(); }
  long long i() const { return (long long)num; }
};

class JsonParser {
 public:
  explicit JsonParser(const std::string &text) : s_(text) {}
  J parse() { J v = value(); ws(); return v; }
 private:
  const std::string &s_;
  size_t p_ = 0;
  void ws() { while (p_ < s_.size() && (s_[p_] == ' ' || s_[p_] == '\n' || s_[p_] == '\t' || s_[p_] == '\r')) ++p_; }
  J value() {
    ws();
    if (p_ >= s_.size()) throw std::runtime_error("json: unexpected end");
    const char c = s_[p_];
    J v;
    if (c == '{') {
      v.t = J::OBJ; ++p_; ws();
      if (s_[p_] == '}') { ++p_; return v; }
      for (;;) {
        ws(); J k = value(); ws();
        if (s_[p_++] != ':') throw std::runtime_error("json: ':' expected");
        v.obj.push_back({k.s, value()}); ws();
        if (s_[p_] == ',') { ++p_; continue; }
        if (s_[p_++] != '}') throw std::runtime_error("json: '}' expected");
        return v;
      }
    }
    if (c == '[') {
      v.t = J::ARR; ++p_; ws();
      if (s_[p_] == ']') { ++p_; return v; }
      for (;;) {
        v.arr.push_back(value()); ws();
        if (s_[p_] == ',') { ++p_; continue; }
        if (s_[p_++] != ']') throw std::runtime_error("json: ']' expected");
        return v;
      }
    }
    if (c == '"') {
      v.t = J::STR; ++p_;
      while (s_[p_] != '"') {
        if (s_[p_] == '\\') {
          ++p_;
          const char e = s_[p_++];
          if (e == 'n') v.s += '\n'; else if (e == 't') v.s += '\t';
          else if (e == 'u') { v.s += (char)std::strtol(s_.substr(p_, 4).c_str(), nullptr, 16); p_ += 4; }
          else v.s += e;
        } else v.s += s_[p_++];
      }
      ++p_;
      return v;
    }
    if (s_.compare(p_, 4, "true") == 0) { v.t = J::BOOL; v.b = true; p_ += 4; return v; }
    if (s_.compare(p_, 5, "false") == 0) { v.t = J::BOOL; v.b = false; p_ += 5; return v; }
    if (s_.compare(p_, 4, "null") == 0) { p_ += 4; return v; }
    char *end = nullptr;
    v.t = J::NUM; v.num = std::strtod(s_.c_str() + p_, &end);
    if (end == s_.c_str() + p_) throw std::runtime_error("json: bad token at " + std::to_string(p_));
    p_ = end - s_.c_str();
    return v;
  }
};

inline J load_json(const std::string &path) {
  std::ifstream f(path);
  if (!f) throw std::runtime_error("cannot open " + path);
  std::stringstream ss; ss << f.rdbuf();
  const std::string text = ss.str();
  return JsonParser(text).parse();
}
