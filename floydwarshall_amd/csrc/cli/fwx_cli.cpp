// fwx_cli -- the reference's command-line loop over the GPU engine.
//
// Mirrors Main.main / userPrompt (/root/reference/src/app/Main.hs:10-37): read a line, serve it
// (rate update first, best-rate query second), print what the reference prints, keep the state.
// At end of input the reference's getLine throws; this loop just stops.
//   usage: fwx_cli [--device N] < session.txt
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
    int device = -1;
    for (int i = 1; i + 1 < argc; ++i)
        if (!strcmp(argv[i], "--device")) device = atoi(argv[i + 1]);
    fwxh_session *s = nullptr;
    if (fwxh_session_create(&s, device) != FWX_OK) {
        fprintf(stderr, "fwx_cli: cannot create session\n");
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
