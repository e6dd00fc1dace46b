// Minimal JSON DOM reader for tokenizer.json (the reference's persistence format,
// tokenizers/src/tokenizer/serialization.rs:14-47).  Read-only, UTF-8, no dependencies.
// Objects keep insertion order (vocab maps can hold >100k keys, so lookups by key are
// linear only for small objects; large maps are iterated).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace tkamd {

struct JsonValue;
using JsonPtr = std::unique_ptr<JsonValue>;

struct JsonValue {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0.0;
    std::string str;
    std::vector<JsonPtr> arr;
    std::vector<std::pair<std::string, JsonPtr>> obj;

    bool is_null() const { return kind == Null; }
    bool is_string() const { return kind == String; }
    bool is_array() const { return kind == Array; }
    bool is_object() const { return kind == Object; }
    bool is_number() const { return kind == Number; }
    bool is_bool() const { return kind == Bool; }

    const JsonValue* get(const char* key) const {
        if (kind != Object) return nullptr;
        for (auto& kv : obj)
            if (kv.first == key) return kv.second.get();
        return nullptr;
    }
    // string member or default
    std::string get_str(const char* key, const std::string& dflt = "") const {
        const JsonValue* v = get(key);
        return (v && v->kind == String) ? v->str : dflt;
    }
    bool get_bool(const char* key, bool dflt) const {
        const JsonValue* v = get(key);
        return (v && v->kind == Bool) ? v->b : dflt;
    }
    double get_num(const char* key, double dflt) const {
        const JsonValue* v = get(key);
        return (v && v->kind == Number) ? v->num : dflt;
    }
};

class JsonParser {
  public:
    JsonParser(const char* s, size_t n) : p_(s), end_(s + n) {}
    JsonPtr parse() {
        JsonPtr v = value();
        ws();
        if (p_ != end_) fail("trailing characters after JSON document");
        return v;
    }

  private:
    const char* p_;
    const char* end_;
    int depth_ = 0;

    [[noreturn]] void fail(const char* msg) { throw std::runtime_error(std::string("tokenizer.json: ") + msg); }
    void ws() {
        while (p_ < end_ && (*p_ == ' ' || *p_ == '\n' || *p_ == '\t' || *p_ == '\r')) ++p_;
    }
    static void put_utf8(std::string& out, uint32_t cp) {
        if (cp < 0x80) out.push_back((char)cp);
        else if (cp < 0x800) { out.push_back((char)(0xC0 | (cp >> 6))); out.push_back((char)(0x80 | (cp & 0x3F))); }
        else if (cp < 0x10000) {
            out.push_back((char)(0xE0 | (cp >> 12))); out.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
            out.push_back((char)(0x80 | (cp & 0x3F)));
        } else {
            out.push_back((char)(0xF0 | (cp >> 18))); out.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
            out.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); out.push_back((char)(0x80 | (cp & 0x3F)));
        }
    }
    uint32_t hex4() {
        if (end_ - p_ < 4) fail("truncated \\u escape");
        uint32_t v = 0;
        for (int i = 0; i < 4; ++i) {
            char c = *p_++;
            v <<= 4;
            if (c >= '0' && c <= '9') v |= (uint32_t)(c - '0');
            else if (c >= 'a' && c <= 'f') v |= (uint32_t)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F') v |= (uint32_t)(c - 'A' + 10);
            else fail("bad hex digit in \\u escape");
        }
        return v;
    }
    std::string string() {
        if (p_ >= end_ || *p_ != '"') fail("expected string");
        ++p_;
        std::string out;
        while (true) {
            if (p_ >= end_) fail("unterminated string");
            // copy a run of plain bytes at once
            const char* q = p_;
            while (q < end_ && *q != '"' && *q != '\\') ++q;
            out.append(p_, q);
            p_ = q;
            if (p_ >= end_) fail("unterminated string");
            if (*p_ == '"') { ++p_; return out; }
            ++p_;  // backslash
            if (p_ >= end_) fail("unterminated escape");
            char c = *p_++;
            switch (c) {
                case '"': out.push_back('"'); break;
                case '\\': out.push_back('\\'); break;
                case '/': out.push_back('/'); break;
                case 'b': out.push_back('\b'); break;
                case 'f': out.push_back('\f'); break;
                case 'n': out.push_back('\n'); break;
                case 'r': out.push_back('\r'); break;
                case 't': out.push_back('\t'); break;
                case 'u': {
                    uint32_t cp = hex4();
                    if (cp >= 0xD800 && cp <= 0xDBFF) {
                        if (end_ - p_ >= 6 && p_[0] == '\\' && p_[1] == 'u') {
                            p_ += 2;
                            uint32_t lo = hex4();
                            if (lo >= 0xDC00 && lo <= 0xDFFF) cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                            else fail("unpaired surrogate in string");
                        } else fail("unpaired surrogate in string");
                    } else if (cp >= 0xDC00 && cp <= 0xDFFF) fail("unpaired surrogate in string");
                    put_utf8(out, cp);
                    break;
                }
                default: fail("bad escape in string");
            }
        }
    }
    JsonPtr value() {
        ws();
        if (p_ >= end_) fail("unexpected end of input");
        if (++depth_ > 256) fail("nesting too deep");
        JsonPtr v(new JsonValue());
        char c = *p_;
        if (c == '{') {
            v->kind = JsonValue::Object;
            ++p_;
            ws();
            if (p_ < end_ && *p_ == '}') { ++p_; --depth_; return v; }
            while (true) {
                ws();
                std::string k = string();
                ws();
                if (p_ >= end_ || *p_ != ':') fail("expected ':'");
                ++p_;
                JsonPtr x = value();
                v->obj.emplace_back(std::move(k), std::move(x));
                ws();
                if (p_ < end_ && *p_ == ',') { ++p_; continue; }
                if (p_ < end_ && *p_ == '}') { ++p_; break; }
                fail("expected ',' or '}'");
            }
        } else if (c == '[') {
            v->kind = JsonValue::Array;
            ++p_;
            ws();
            if (p_ < end_ && *p_ == ']') { ++p_; --depth_; return v; }
            while (true) {
                v->arr.push_back(value());
                ws();
                if (p_ < end_ && *p_ == ',') { ++p_; continue; }
                if (p_ < end_ && *p_ == ']') { ++p_; break; }
                fail("expected ',' or ']'");
            }
        } else if (c == '"') {
            v->kind = JsonValue::String;
            v->str = string();
        } else if (c == 't' && end_ - p_ >= 4 && !memcmp(p_, "true", 4)) {
            v->kind = JsonValue::Bool; v->b = true; p_ += 4;
        } else if (c == 'f' && end_ - p_ >= 5 && !memcmp(p_, "false", 5)) {
            v->kind = JsonValue::Bool; v->b = false; p_ += 5;
        } else if (c == 'n' && end_ - p_ >= 4 && !memcmp(p_, "null", 4)) {
            v->kind = JsonValue::Null; p_ += 4;
        } else if (c == '-' || (c >= '0' && c <= '9')) {
            const char* q = p_;
            if (*q == '-') ++q;
            while (q < end_ && ((*q >= '0' && *q <= '9') || *q == '.' || *q == 'e' || *q == 'E' || *q == '+' || *q == '-')) ++q;
            std::string t(p_, q);
            char* e = nullptr;
            v->kind = JsonValue::Number;
            v->num = strtod(t.c_str(), &e);
            if (e == t.c_str()) fail("bad number");
            p_ = q;
        } else {
            fail("unexpected character");
        }
        --depth_;
        return v;
    }
};

inline JsonPtr json_parse(const char* s, size_t n) { return JsonParser(s, n).parse(); }

}  // namespace tkamd
