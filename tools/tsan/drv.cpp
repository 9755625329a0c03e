#include <cstdio>
#include <string>
#include "../../include/rtx.h"
#include "../../include/rtx_host.h"
#include "../../rust-raytracer_amd/csrc/bvh_build.h"
#include "../../rust-raytracer_amd/csrc/flatten.h"
int main() {
    for (const char *name : {"scan", "basic_suzanne"}) {
        rtx_host_scene *s = rtx_scene_builtin(name, 256, ASSETS_DIR, name[0] == 's' ? 300000 : 0);
        if (!s) { std::printf("scene failed: %s\n", rtx_host_last_error()); return 1; }
        const rtx_scene_desc *d = rtx_scene_get_desc(s);
        rtx::BuiltBvh bvh; std::string err;
        int rc = rtx::build_bvh(*d, 8, bvh, err);
        rtx::FlatScene fs;
        int rc2 = rtx::flatten_scene(*d, bvh, fs, err, 8);
        std::printf("%s: build %d flatten %d nodes %zu wide %zu prims %zu\n", name, rc, rc2, fs.nodes.size(), fs.wnodes.size(), fs.prims.size());
        rtx_scene_free(s);
    }
    return 0;
}
