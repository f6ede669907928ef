// fwx_cli -- the reference's command-line loop over the GPU engine.
//
// Mirrors Main.main / userPrompt (/root/reference/src/app/Main.hs:10-37): read a line, serve it
// (rate update first, best-rate query second), print what the reference prints, keep the state.
// At end of input the reference's getLine throws; this loop just stops.
//   usage: fwx_cli [--device N] [--devices 0,1,...,7 [--multi-from V]] < session.txt
//   --devices: row-partition the solved matrix over these GPUs (one process, the whole node behind
//   the one floydWarshall call) from V vertices on (default 4096); a device may repeat.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "fwx.h"
#include "fwx_host.h"

int main(int argc, char **argv)
{
    int device = -1, multi_from = 4096;
    std::vector<int32_t> devices;
    for (int i = 1; i + 1 < argc; ++i) {
        if (!strcmp(argv[i], "--device")) device = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "--multi-from")) multi_from = atoi(argv[i + 1]);
        if (!strcmp(argv[i], "--devices"))
            for (const char *p = argv[i + 1]; *p;) {
                devices.push_back((int32_t)strtol(p, const_cast<char **>(&p), 10));
                if (*p == ',') ++p;
            }
    }
    fwxh_session *s = nullptr;
    if (fwxh_session_create(&s, device) != FWX_OK) {
        fprintf(stderr, "fwx_cli: cannot create session\n");
        return 1;
    }
    if (!devices.empty() &&
        fwxh_session_set_devices(s, (int32_t)devices.size(), devices.data(), multi_from) != FWX_OK) {
        fprintf(stderr, "fwx_cli: bad --devices list\n");
        return 1;
    }
    std::string line;
    std::vector<char> out(1 << 16);
    while (std::getline(std::cin, line)) {
        int rc;
        while ((rc = fwxh_serve_line(s, line.c_str(), out.data(), out.size())) == FWX_ERR_CAPACITY)
            out.resize(out.size() * 4);
        if (rc < 0) {
            fprintf(stderr, "fwx_cli: %s\n", fwx_strerror(rc));
            fwxh_session_destroy(s);
            return 2;
        }
        fputs(out.data(), stdout);
        fflush(stdout);
    }
    fwxh_session_destroy(s);
    return 0;
}
