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
        // the same tree with its top split through the TopSplitter interface and the pieces built by a pool of 8 threads
        rtx::BuiltBvh bvh2;
        rtx::TopSplitter *top = rtx::new_host_top_splitter();
        int rc3 = rtx::build_bvh(*d, 8, bvh2, err, top, 512u);
        delete top;
        bool same = rc3 == rc && bvh2.leaf_order == bvh.leaf_order && bvh2.depth == bvh.depth && bvh2.nodes.size() == bvh.nodes.size();
        std::printf("%s: top phase + pieces %d, same tree: %s\n", name, rc3, same ? "yes" : "NO");
        if (!same) return 1;
        rtx_scene_free(s);
    }
    return 0;
}
