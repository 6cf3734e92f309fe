#include "json_out.hpp"

#include <cstdio>

namespace lashhost {

std::string json_escape(const std::string &s)
{
    std::string o = "\"";
    for (unsigned char c : s) {
        switch (c) {
        case '"': o += "\\\""; break;
        case '\\': o += "\\\\"; break;
        case '\b': o += "\\b"; break;
        case '\f': o += "\\f"; break;
        case '\n': o += "\\n"; break;
        case '\r': o += "\\r"; break;
        case '\t': o += "\\t"; break;
        default:
            if (c < 0x20) { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; }
            else o += (char)c;
        }
    }
    return o + "\"";
}

std::string json_pretty_string_array(const std::vector<std::string> &items)
{
    if (items.empty()) return "[]";
    std::string o = "[\n";
    for (size_t i = 0; i < items.size(); ++i) {
        o += "  " + json_escape(items[i]);
        o += (i + 1 < items.size()) ? ",\n" : "\n";
    }
    return o + "]";
}

std::string json_pretty_string_object(const std::map<std::string, std::string> &kv)
{
    if (kv.empty()) return "{}";
    std::string o = "{\n";
    size_t i = 0;
    for (const auto &e : kv) {
        o += "  " + json_escape(e.first) + ": " + json_escape(e.second);
        o += (++i < kv.size()) ? ",\n" : "\n";
    }
    return o + "}";
}

namespace {
struct Cur { const std::string &t; size_t i = 0; };
void ws(Cur &c) { while (c.i < c.t.size() && (c.t[c.i] == ' ' || c.t[c.i] == '\n' || c.t[c.i] == '\r' || c.t[c.i] == '\t')) ++c.i; }
bool str(Cur &c, std::string &out)
{
    ws(c);
    if (c.i >= c.t.size() || c.t[c.i] != '"') return false;
    ++c.i;
    out.clear();
    while (c.i < c.t.size() && c.t[c.i] != '"') {
        char ch = c.t[c.i++];
        if (ch != '\\') { out += ch; continue; }
        if (c.i >= c.t.size()) return false;
        char e = c.t[c.i++];
        switch (e) {
        case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break;
        case 'b': out += '\b'; break; case 'f': out += '\f'; break; case '/': out += '/'; break;
        case '"': out += '"'; break; case '\\': out += '\\'; break;
        case 'u': {
            if (c.i + 4 > c.t.size()) return false;
            unsigned v = 0;
            for (int k = 0; k < 4; ++k) {
                char h = c.t[c.i++];
                v = v * 16 + (h >= '0' && h <= '9' ? h - '0' : (h | 32) >= 'a' && (h | 32) <= 'f' ? (h | 32) - 'a' + 10 : 0);
            }
            if (v < 0x80) out += (char)v;
            else if (v < 0x800) { out += (char)(0xC0 | (v >> 6)); out += (char)(0x80 | (v & 0x3F)); }
            else { out += (char)(0xE0 | (v >> 12)); out += (char)(0x80 | ((v >> 6) & 0x3F)); out += (char)(0x80 | (v & 0x3F)); }
            break;
        }
        default: return false;
        }
    }
    if (c.i >= c.t.size()) return false;
    ++c.i;
    return true;
}
}  // namespace

bool json_parse_string_array(const std::string &text, std::vector<std::string> &out)
{
    Cur c{text};
    out.clear();
    ws(c);
    if (c.i >= text.size() || text[c.i] != '[') return false;
    ++c.i;
    ws(c);
    if (c.i < text.size() && text[c.i] == ']') return true;
    for (;;) {
        std::string s;
        if (!str(c, s)) return false;
        out.push_back(s);
        ws(c);
        if (c.i < text.size() && text[c.i] == ',') { ++c.i; continue; }
        if (c.i < text.size() && text[c.i] == ']') return true;
        return false;
    }
}

bool json_parse_string_object(const std::string &text, std::map<std::string, std::string> &out)
{
    Cur c{text};
    out.clear();
    ws(c);
    if (c.i >= text.size() || text[c.i] != '{') return false;
    ++c.i;
    ws(c);
    if (c.i < text.size() && text[c.i] == '}') return true;
    for (;;) {
        std::string k, v;
        if (!str(c, k)) return false;
        ws(c);
        if (c.i >= text.size() || text[c.i] != ':') return false;
        ++c.i;
        if (!str(c, v)) return false;
        out[k] = v;
        ws(c);
        if (c.i < text.size() && text[c.i] == ',') { ++c.i; continue; }
        if (c.i < text.size() && text[c.i] == '}') return true;
        return false;
    }
}

}  // namespace lashhost
