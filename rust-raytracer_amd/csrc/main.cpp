// Command-line front end with the reference's flags (src/main.rs:26-77):
//   -f/--file out.png  -t/--threads 4  -s/--samples_per_pixel 1  -b/--max_ray_bounces 4
//   -w/--pixel_width 500
// plus --scene NAME (the reference picks its scene by editing main.rs:82-92; its
// checked-in choice, load_random_spheres(1000), is the default), --media DIR, --device N.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "../../include/rtx_host.h"

int main(int argc, char **argv) {
    std::string file = "out.png", scene = "random_spheres", media = "assets";
    unsigned threads = 4, spp = 1, bounces = 4, width = 500;
    unsigned long long param = 0;
    int device = 0;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&](const char *name) -> const char * {
            if (i + 1 >= argc) {
                std::fprintf(stderr, "error: %s needs a value\n", name);
                std::exit(2);
            }
            return argv[++i];
        };
        if (a == "-f" || a == "--file") file = val("-f");
        else if (a == "-t" || a == "--threads") threads = (unsigned)std::strtoul(val("-t"), nullptr, 10);
        else if (a == "-s" || a == "--samples_per_pixel") spp = (unsigned)std::strtoul(val("-s"), nullptr, 10);
        else if (a == "-b" || a == "--max_ray_bounces") bounces = (unsigned)std::strtoul(val("-b"), nullptr, 10);
        else if (a == "-w" || a == "--pixel_width") width = (unsigned)std::strtoul(val("-w"), nullptr, 10);
        else if (a == "--scene") scene = val("--scene");
        else if (a == "--media") media = val("--media");
        else if (a == "--param") param = std::strtoull(val("--param"), nullptr, 10);
        else if (a == "--device") device = std::atoi(val("--device"));
        else if (a == "-h" || a == "--help") {
            std::printf("Rust Ray Tracer (MI355X trace backend) 0.1\n"
                        "  -f, --file FILE              File to save rendered image as .png [out.png]\n"
                        "  -t, --threads N              Number of host threads (BVH build) [4]\n"
                        "  -s, --samples_per_pixel N    Number of rays to cast per pixel [1]\n"
                        "  -b, --max_ray_bounces N      Max number of times to bounce a ray [4]\n"
                        "  -w, --pixel_width N          Width of the outputted image in pixels [500]\n"
                        "      --scene NAME             basic|random_spheres|suzanne|basic_suzanne|scan|scan_glass|soup|mesh:FILE.obj|mesh:FILE.ply|file:FILE.scene\n"
                        "      --media DIR, --param N, --device N\n");
            return 0;
        } else {
            std::fprintf(stderr, "error: unknown argument '%s'\n", a.c_str());
            return 2;
        }
    }
    int rc = rtx_render_builtin_to_png(scene.c_str(), width, media.c_str(), param, spp, bounces, file.c_str(), threads, device);
    if (rc != RTX_OK) {
        std::fprintf(stderr, "error %d: %s\n", rc, rtx_host_last_error());
        return 1;
    }
    return 0;
}
