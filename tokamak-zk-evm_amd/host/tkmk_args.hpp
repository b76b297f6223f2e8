// tkmk_args.hpp — the argument surface of the three backend binaries as `tokamak-cli` drives them, and the subcircuit-library
// resolution of the reference's release builds.
//
// What the CLI sends (packages/cli/src/cli.ts:537-546 `backendOutputArgs`, runtime.ts:1840-1848):
//     preprocess | prove   --crs DIR --synthesizer-stat DIR --output DIR
//     trusted-setup        --output DIR --fixed-tau
// i.e. NO --subcircuit-library: the reference's release binaries embed the library and materialise it under
// <cache>/tokamak-zk-evm/subcircuit-library/<snapshot>/library (libs/src/subcircuit_library.rs:41-122); only non-release
// builds take the flag (`SubcircuitLibraryArg`, :17-23) and panic without it (:55-57).  The binaries here accept the flag
// and, when it is absent, look for the library where a tokamak-cli installation has it (resolve_subcircuit_library below).
// Flags are parsed the way clap does for `#[arg(long)]` options: `--flag value` and `--flag=value`, any order, each at most once.
#pragma once
#include <algorithm>
#include <dirent.h>
#include <sys/stat.h>
#include <unistd.h>

#include <climits>
#include <cstdlib>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "tkmk_host.hpp"

namespace tkmk {
namespace args {

struct Spec {
    std::set<std::string> valued;   // "--crs", ...
    std::set<std::string> switches; // "--fixed-tau", ...
};

// `--version` / `-V` (clap's `#[command(version)]`): "<binary> <version>".  `tokamak-cli doctor` runs every backend binary with it and
// reads the first x.y.z it finds (packages/cli/src/cli.ts:390-404, 655-664).  The number is the version of the reference's backend
// workspace whose command-line surface and file formats these binaries mirror (packages/backend/Cargo.toml:13), with a build tag.
#define TKMK_BACKEND_INTERFACE_VERSION "2.1.3+mi355x"

struct Parsed {
    std::map<std::string, std::string> values;
    std::set<std::string> switches;
    bool help = false, version = false;
    std::string error;   // non-empty: print it + usage, exit 2 (clap's exit code for usage errors)
    bool has(const std::string &k) const { return values.count(k) != 0; }
    std::string get(const std::string &k, const std::string &dflt = "") const {
        auto it = values.find(k);
        return it == values.end() ? dflt : it->second;
    }
    bool flag(const std::string &k) const { return switches.count(k) != 0; }
};

inline Parsed parse(int argc, char **argv, const Spec &spec) {
    Parsed p;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        if (a == "--help" || a == "-h") {
            p.help = true;
            continue;
        }
        if (a == "--version" || a == "-V") {
            p.version = true;
            continue;
        }
        std::string key = a, val;
        bool inline_value = false;
        size_t eq = a.find('=');
        if (a.rfind("--", 0) == 0 && eq != std::string::npos) key = a.substr(0, eq), val = a.substr(eq + 1), inline_value = true;
        if (spec.switches.count(key)) {
            if (inline_value) {
                p.error = "unexpected value for '" + key + "'";
                return p;
            }
            p.switches.insert(key);
        } else if (spec.valued.count(key)) {
            if (!inline_value) {
                if (i + 1 >= argc) {
                    p.error = "a value is required for '" + key + " <PATH>' but none was supplied";
                    return p;
                }
                val = argv[++i];
            }
            if (p.values.count(key)) {
                p.error = "the argument '" + key + " <PATH>' cannot be used multiple times";
                return p;
            }
            p.values[key] = val;
        } else {
            p.error = "unexpected argument '" + a + "' found";
            return p;
        }
    }
    return p;
}

inline bool is_library_dir(const std::string &d) {
    struct stat st;
    return !d.empty() && ::stat((d + "/setupParams.json").c_str(), &st) == 0 && S_ISREG(st.st_mode);
}

inline std::string exe_dir() {
    char buf[PATH_MAX];
    ssize_t n = ::readlink("/proc/self/exe", buf, sizeof buf - 1);
    if (n <= 0) return "";
    buf[n] = 0;
    std::string p = buf;
    size_t s = p.rfind('/');
    return s == std::string::npos ? "" : p.substr(0, s);
}

// cache_root_dir() of the reference (libs/src/subcircuit_library.rs:112-128), Linux branch — WITHOUT its last two fallbacks ($TMPDIR, /tmp):
// the reference opens exactly the snapshot named by the integrity hash it embeds, so a world-writable directory is harmless there; this
// resolver has no such hash and must not pick up a directory anybody on the machine may have written.  "" = no per-user cache directory.
inline std::string cache_root_dir() {
    if (const char *x = std::getenv("XDG_CACHE_HOME"); x && *x) return x;
    if (const char *h = std::getenv("HOME"); h && *h) return std::string(h) + "/.cache";
    return "";
}

/* --subcircuit-library given: that directory (canonicalised; the reference panics "cannot resolve subcircuit library path {path}",
 * subcircuit_library.rs:42-45).  Absent — the case under tokamak-cli — in this order:
 *   1. $TKMK_SUBCIRCUIT_LIBRARY
 *   2. next to the installation: <exe>/../resource/qap-compiler/library, <exe>/../subcircuit-library, <exe>/../library
 *      (tokamak-cli keeps binaries under <runtime>/bin and resources under <runtime>/resource, runtime.ts:465-485)
 *   3. the directory a reference release binary of the same installation has materialised:
 *      <cache>/tokamak-zk-evm/subcircuit-library/<snapshot>/library, <cache> = $XDG_CACHE_HOME or $HOME/.cache only — if exactly ONE
 *      snapshot holds a setupParams.json; several snapshots are an error that lists them (no guessing by modification time)
 * The binaries print the directory they resolved ("Subcircuit library: ...") before they read it.
 * Nothing found: the reference's message for a binary without an embedded library, plus where this one looked. */
inline std::string resolve_subcircuit_library(const Parsed &p) {
    if (p.has("--subcircuit-library")) {
        std::string given = p.get("--subcircuit-library");
        char real[PATH_MAX];
        if (!::realpath(given.c_str(), real)) throw Error("cannot resolve subcircuit library path " + given);
        return real;
    }
    std::vector<std::string> tried;
    auto ok = [&](const std::string &d) {
        if (d.empty()) return false;
        tried.push_back(d);
        return is_library_dir(d);
    };
    if (const char *e = std::getenv("TKMK_SUBCIRCUIT_LIBRARY"); e && *e) {
        if (ok(e)) return e;
        throw Error(std::string("TKMK_SUBCIRCUIT_LIBRARY=") + e + " holds no setupParams.json");
    }
    std::string exe = exe_dir();
    if (!exe.empty())
        for (const char *rel : {"/../resource/qap-compiler/library", "/../subcircuit-library", "/../library"})
            if (ok(exe + rel)) {   // canonical, so that the "Subcircuit library:" line names the directory without a /bin/.. detour
                char real[PATH_MAX];
                return ::realpath((exe + rel).c_str(), real) ? std::string(real) : exe + rel;
            }
    const std::string cache = cache_root_dir();
    if (!cache.empty()) {
        std::string snapshots = cache + "/tokamak-zk-evm/subcircuit-library";
        tried.push_back(snapshots + "/*/library");
        std::vector<std::string> found;
        if (DIR *d = ::opendir(snapshots.c_str())) {
            while (dirent *e = ::readdir(d)) {
                std::string name = e->d_name;
                if (name == "." || name == ".." || name.rfind("staging-", 0) == 0) continue;
                std::string lib = snapshots + "/" + name + "/library";
                if (is_library_dir(lib)) found.push_back(lib);
            }
            ::closedir(d);
        }
        if (found.size() == 1) return found[0];
        if (found.size() > 1) {
            // the reference would open the one whose name is its embedded integrity hash; without that hash any choice is a guess, and a
            // wrong guess is a CRS or a proof for another circuit
            std::sort(found.begin(), found.end());
            std::string msg = "more than one subcircuit-library snapshot found, refusing to pick one:";
            for (const std::string &f : found) msg += " " + f + ";";
            msg += " set TKMK_SUBCIRCUIT_LIBRARY or pass --subcircuit-library";
            throw Error(msg);
        }
    }
    std::string msg = "--subcircuit-library is required: no subcircuit library found (looked in";
    for (const std::string &t : tried) msg += " " + t + ";";
    msg += " set TKMK_SUBCIRCUIT_LIBRARY or pass the flag)";
    throw Error(msg);
}

}  // namespace args
}  // namespace tkmk
