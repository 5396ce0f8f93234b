// Compile DIR/unit.hip (+ the headers in DIR) with hipRTC exactly as csrc/mm_rtc.hip does (same options), write the code object:
//   g++ -O1 -o /tmp/hiprtc_compile tools/experiments/hiprtc_compile.cpp -ldl
//   /tmp/hiprtc_compile DIR gfx950:sramecc+:xnack- DIM OUT.hsaco [libhiprtc path] [extra options ...]
// No GPU is needed (the architecture is given).  Used to compare hipRTC's code object with `hipcc --genco`'s for the unit whose
// lanes-in-step NUTS kernel hipRTC gets wrong (DESIGN_LOG 13.5.5).
#include <dirent.h>
#include <dlfcn.h>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <vector>
typedef void *prog_t;
static std::string slurp(const std::string &p)
{
    std::ifstream f(p, std::ios::binary);
    std::stringstream ss;
    ss << f.rdbuf();
    return ss.str();
}
int main(int argc, char **argv)
{
    if (argc < 5)
        return 2;
    const std::string dir = argv[1], arch = argv[2], dim = argv[3], out = argv[4];
    const char *libp = argc > 5 ? argv[5] : "/opt/rocm/lib/libhiprtc.so";
    void *h = dlopen(libp, RTLD_NOW | RTLD_LOCAL);
    if (!h) {
        fprintf(stderr, "dlopen: %s\n", dlerror());
        return 1;
    }
    auto Create = (int (*)(prog_t *, const char *, const char *, int, const char **, const char **))dlsym(h, "hiprtcCreateProgram");
    auto Compile = (int (*)(prog_t, int, const char **))dlsym(h, "hiprtcCompileProgram");
    auto LogSize = (int (*)(prog_t, size_t *))dlsym(h, "hiprtcGetProgramLogSize");
    auto Log = (int (*)(prog_t, char *))dlsym(h, "hiprtcGetProgramLog");
    auto CodeSize = (int (*)(prog_t, size_t *))dlsym(h, "hiprtcGetCodeSize");
    auto Code = (int (*)(prog_t, char *))dlsym(h, "hiprtcGetCode");
    std::vector<std::string> names, texts;
    DIR *d = opendir(dir.c_str());
    while (dirent *e = readdir(d)) {
        const std::string n = e->d_name;
        if (n.size() > 2 && (n.substr(n.size() - 2) == ".h" || (n.size() > 4 && n.substr(n.size() - 4) == ".inc"))) {
            names.push_back(n);
            texts.push_back(slurp(dir + "/" + n));
        }
    }
    closedir(d);
    std::vector<const char *> hn, ht;
    for (size_t i = 0; i < names.size(); ++i) {
        hn.push_back(names[i].c_str());
        ht.push_back(texts[i].c_str());
    }
    const std::string src = slurp(dir + "/unit.hip");
    prog_t p = nullptr;
    if (Create(&p, src.c_str(), "unit.hip", (int)names.size(), ht.data(), hn.data()) != 0)
        return 1;
    const std::string a = "--offload-arch=" + arch, dd = "-DMM_USER_DIM=" + dim;
    std::vector<const char *> opts = {a.c_str(), "-O3", "-ffp-contract=off", "-std=c++17", dd.c_str(), "-Wno-pass-failed"};
    for (int i = 6; i < argc; ++i)
        opts.push_back(argv[i]);
    const int rc = Compile(p, (int)opts.size(), opts.data());
    size_t ls = 0;
    if (LogSize(p, &ls) == 0 && ls > 1) {
        std::vector<char> l(ls + 1, 0);
        Log(p, l.data());
        fprintf(stderr, "%s\n", l.data());
    }
    if (rc != 0) {
        fprintf(stderr, "compile failed: %d\n", rc);
        return 1;
    }
    size_t cs = 0;
    CodeSize(p, &cs);
    std::vector<char> c(cs);
    Code(p, c.data());
    FILE *f = fopen(out.c_str(), "wb");
    fwrite(c.data(), 1, cs, f);
    fclose(f);
    printf("%zu bytes\n", cs);
    return 0;
}
