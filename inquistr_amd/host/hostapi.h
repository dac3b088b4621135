// hostapi.h — the CLI's way into libinquistr_host.so: loaded when a command needs it, not when the process starts.
// `inquistr call` with INQ_SERVER set only talks to a server over a socket; binding the executable to the library would make
// every such process map the HIP runtime and its dozen dependencies first (12 ms of a 75 ms call, measured), for nothing.
#pragma once
#include <dlfcn.h>
#include <unistd.h>

#include <climits>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "../../include/inquistr_host.h"

namespace inq {

struct HostApi {
    decltype(&::inq_genotype_repeats) genotype_repeats;
    decltype(&::inq_genotype_repeats_devices) genotype_repeats_devices;
    decltype(&::inq_host_ctx_option) ctx_option;
    decltype(&::inq_combine) combine;
    decltype(&::inq_outlier) outlier;
    decltype(&::inq_host_sample_name) host_sample_name;
    decltype(&::inq_session_open) session_open;
    decltype(&::inq_session_call_many) session_call_many;
    decltype(&::inq_session_stage) session_stage;
    decltype(&::inq_session_run) session_run;
    decltype(&::inq_session_discard) session_discard;
    decltype(&::inq_session_close) session_close;
};

// the library next to the executable (or INQ_HOST_LIB, as the Python binding reads it); a process that cannot load it ends here
inline const HostApi &host_api() {
    static const HostApi api = [] {
        std::string path;
        if (const char *e = std::getenv("INQ_HOST_LIB"); e && *e) path = e;
        else {
            char exe[PATH_MAX];
            const ssize_t n = ::readlink("/proc/self/exe", exe, sizeof exe - 1);
            path = n > 0 ? std::string(exe, (size_t)n) : std::string("./inquistr");
            path = path.substr(0, path.rfind('/') + 1) + "libinquistr_host.so";
        }
        void *h = ::dlopen(path.c_str(), RTLD_NOW | RTLD_GLOBAL);
        if (!h) {
            std::fprintf(stderr, "cannot load %s: %s\n", path.c_str(), ::dlerror());
            std::_Exit(1);
        }
        HostApi a;
        auto sym = [&](const char *name) {
            void *p = ::dlsym(h, name);
            if (!p) {
                std::fprintf(stderr, "%s lacks %s\n", path.c_str(), name);
                std::_Exit(1);
            }
            return p;
        };
        a.genotype_repeats = reinterpret_cast<decltype(a.genotype_repeats)>(sym("inq_genotype_repeats"));
        a.genotype_repeats_devices = reinterpret_cast<decltype(a.genotype_repeats_devices)>(sym("inq_genotype_repeats_devices"));
        a.ctx_option = reinterpret_cast<decltype(a.ctx_option)>(sym("inq_host_ctx_option"));
        a.combine = reinterpret_cast<decltype(a.combine)>(sym("inq_combine"));
        a.outlier = reinterpret_cast<decltype(a.outlier)>(sym("inq_outlier"));
        a.host_sample_name = reinterpret_cast<decltype(a.host_sample_name)>(sym("inq_host_sample_name"));
        a.session_open = reinterpret_cast<decltype(a.session_open)>(sym("inq_session_open"));
        a.session_call_many = reinterpret_cast<decltype(a.session_call_many)>(sym("inq_session_call_many"));
        a.session_stage = reinterpret_cast<decltype(a.session_stage)>(sym("inq_session_stage"));
        a.session_run = reinterpret_cast<decltype(a.session_run)>(sym("inq_session_run"));
        a.session_discard = reinterpret_cast<decltype(a.session_discard)>(sym("inq_session_discard"));
        a.session_close = reinterpret_cast<decltype(a.session_close)>(sym("inq_session_close"));
        return a;
    }();
    return api;
}

}  // namespace inq
