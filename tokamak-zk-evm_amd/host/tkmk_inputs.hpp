// tkmk_inputs.hpp — readers of the synthesizer documents `prove` takes (packages/backend/libs/src/iotools/mod.rs:366-372 PlacementVariables,
// :399-406 Instance): hex-string lists -> ScalarField through ScalarField::from_hex (:126-146).
#pragma once
#include <fstream>
#include <string>
#include <vector>

#include "tkmk_fr.hpp"
#include "tkmk_json.hpp"
#include "tkmk_protocol.hpp"

namespace tkmk {

inline std::vector<ScalarField> hex_list(const json::Value &arr) {
    std::vector<ScalarField> out;
    out.reserve(arr.items().size());
    for (const json::Value &e : arr.items()) out.push_back(fr_from_hex(e.as_string()));
    return out;
}

// placementVariables.json is the one large input (tens of MB of hex text: [{"subcircuitId": N, "variables": ["0x..", ...]}, ...]); it is
// scanned directly — every hex string goes straight to ScalarField::from_hex — instead of through the DOM of tkmk_json.hpp.
// Accepts any key order and whitespace; any other key's value must be a number or a string.
inline std::vector<PlacementVariables> parse_placement_variables(const std::string &t) {
    size_t i = 0;
    const size_t n = t.size();
    auto fail = [&](const char *what) { throw Error(std::string("placementVariables.json: ") + what + " at byte " + std::to_string(i)); };
    auto ws = [&]() { while (i < n && (t[i] == ' ' || t[i] == '\n' || t[i] == '\r' || t[i] == '\t')) i++; };
    auto expect = [&](char c) { ws(); if (i >= n || t[i] != c) fail("unexpected character"); i++; };
    auto str = [&](size_t &b, size_t &e) {   // plain strings only (no escapes occur in keys or hex text)
        expect('"');
        b = i;
        while (i < n && t[i] != '"') { if (t[i] == '\\') fail("escape in string"); i++; }
        if (i >= n) fail("unterminated string");
        e = i++;
    };
    std::vector<PlacementVariables> out;
    expect('[');
    ws();
    if (i < n && t[i] == ']') return out;
    for (;;) {
        expect('{');
        PlacementVariables pl{};
        bool has_id = false, has_vars = false;
        for (;;) {
            size_t kb, ke;
            str(kb, ke);
            expect(':');
            ws();
            std::string key = t.substr(kb, ke - kb);
            if (key == "subcircuitId") {
                size_t b = i;
                while (i < n && t[i] >= '0' && t[i] <= '9') i++;
                if (b == i) fail("expected a non-negative integer");
                pl.subcircuitId = (size_t)std::stoull(t.substr(b, i - b));
                has_id = true;
            } else if (key == "variables") {
                expect('[');
                ws();
                if (i < n && t[i] == ']') i++;
                else
                    for (;;) {
                        size_t b, e;
                        str(b, e);
                        pl.variables.push_back(fr_from_hex(t.data() + b, e - b));
                        ws();
                        if (i < n && t[i] == ',') { i++; continue; }
                        expect(']');
                        break;
                    }
                has_vars = true;
            } else if (i < n && t[i] == '"') {
                size_t b, e;
                str(b, e);
            } else {
                while (i < n && t[i] != ',' && t[i] != '}') i++;
            }
            ws();
            if (i < n && t[i] == ',') { i++; continue; }
            expect('}');
            break;
        }
        if (!has_id || !has_vars) fail("placement without subcircuitId / variables");
        out.push_back(std::move(pl));
        ws();
        if (i < n && t[i] == ',') { i++; continue; }
        expect(']');
        break;
    }
    return out;
}

inline std::vector<PlacementVariables> read_placement_variables(const std::string &path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) throw Error("cannot open " + path);
    std::string t((size_t)f.tellg(), '\0');
    f.seekg(0);
    f.read(&t[0], (std::streamsize)t.size());
    return parse_placement_variables(t);
}

}  // namespace tkmk
