// tkmk_json.hpp — a small JSON reader for the prover's input files (setupParams.json, permutation.json, instance.json,
// placementVariables.json, subcircuitInfo.json): objects, arrays, strings, non-negative integers, true / false / null.
// The reference uses serde_json (libs/src/iotools/mod.rs:30-120); this is the dependency-free stand-in for the C++ host side.
#pragma once
#include <cctype>
#include <fstream>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace tkmk {
namespace json {

struct Value {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0;
    std::string str;   // String; for Number: the literal text (exact for integers)
    std::vector<Value> arr;
    std::vector<std::pair<std::string, Value>> obj;

    const Value &at(const std::string &key) const {
        if (kind != Object) throw std::runtime_error("json: not an object (looking for '" + key + "')");
        for (auto &kv : obj)
            if (kv.first == key) return kv.second;
        throw std::runtime_error("json: missing field '" + key + "'");
    }
    bool has(const std::string &key) const {
        if (kind != Object) return false;
        for (auto &kv : obj)
            if (kv.first == key) return true;
        return false;
    }
    const std::vector<Value> &items() const {
        if (kind != Array) throw std::runtime_error("json: not an array");
        return arr;
    }
    size_t as_size() const {
        if (kind != Number || str.empty() || str.find_first_not_of("0123456789") != std::string::npos)
            throw std::runtime_error("json: expected a non-negative integer");
        return (size_t)std::stoull(str);
    }
    const std::string &as_string() const {
        if (kind != String) throw std::runtime_error("json: expected a string");
        return str;
    }
};

class Parser {
    const std::string &s;
    size_t i = 0;
    void ws() {
        while (i < s.size() && std::isspace((unsigned char)s[i])) i++;
    }
    [[noreturn]] void fail(const char *m) { throw std::runtime_error(std::string("json: ") + m + " at offset " + std::to_string(i)); }
    std::string string_() {
        if (s[i] != '"') fail("expected string");
        i++;
        std::string out;
        while (i < s.size() && s[i] != '"') {
            char c = s[i++];
            if (c == '\\') {
                if (i >= s.size()) fail("bad escape");
                char e = s[i++];
                switch (e) {
                    case 'n': out.push_back('\n'); break;
                    case 't': out.push_back('\t'); break;
                    case 'r': out.push_back('\r'); break;
                    case 'b': out.push_back('\b'); break;
                    case 'f': out.push_back('\f'); break;
                    case 'u':   // the prover's files are ASCII; keep the code unit's low byte
                        if (i + 4 > s.size()) fail("bad \\u escape");
                        out.push_back((char)std::stoi(s.substr(i + 2, 2), nullptr, 16));
                        i += 4;
                        break;
                    default: out.push_back(e);
                }
            } else {
                out.push_back(c);
            }
        }
        if (i >= s.size()) fail("unterminated string");
        i++;
        return out;
    }
    Value value() {
        ws();
        if (i >= s.size()) fail("unexpected end");
        Value v;
        char c = s[i];
        if (c == '{') {
            v.kind = Value::Object;
            i++;
            ws();
            if (s[i] == '}') { i++; return v; }
            for (;;) {
                ws();
                std::string k = string_();
                ws();
                if (s[i] != ':') fail("expected ':'");
                i++;
                v.obj.emplace_back(std::move(k), value());
                ws();
                if (s[i] == ',') { i++; continue; }
                if (s[i] == '}') { i++; break; }
                fail("expected ',' or '}'");
            }
        } else if (c == '[') {
            v.kind = Value::Array;
            i++;
            ws();
            if (s[i] == ']') { i++; return v; }
            for (;;) {
                v.arr.push_back(value());
                ws();
                if (s[i] == ',') { i++; continue; }
                if (s[i] == ']') { i++; break; }
                fail("expected ',' or ']'");
            }
        } else if (c == '"') {
            v.kind = Value::String;
            v.str = string_();
        } else if (c == 't' && s.compare(i, 4, "true") == 0) {
            v.kind = Value::Bool, v.b = true, i += 4;
        } else if (c == 'f' && s.compare(i, 5, "false") == 0) {
            v.kind = Value::Bool, i += 5;
        } else if (c == 'n' && s.compare(i, 4, "null") == 0) {
            i += 4;
        } else {
            size_t j = i;
            while (j < s.size() && (std::isdigit((unsigned char)s[j]) || s[j] == '-' || s[j] == '+' || s[j] == '.' || s[j] == 'e' || s[j] == 'E')) j++;
            if (j == i) fail("unexpected character");
            v.kind = Value::Number;
            v.str = s.substr(i, j - i);
            v.num = std::stod(v.str);
            i = j;
        }
        return v;
    }

  public:
    explicit Parser(const std::string &text) : s(text) {}
    Value parse() {
        Value v = value();
        ws();
        if (i != s.size()) fail("trailing characters");
        return v;
    }
};

inline Value parse(const std::string &text) { return Parser(text).parse(); }
inline Value read_file(const std::string &path) {
    std::ifstream f(path);
    if (!f) throw std::runtime_error("cannot open " + path);
    std::string text((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    return parse(text);
}

}  // namespace json
}  // namespace tkmk
