// dropin_demo.cpp -- TEST INFRASTRUCTURE: a program written the way a user of the reference
// writes one (scene built with the reference's own classes, Scene::take, buildBVH, Camera),
// whose pixel loop is replaced by the drop-in call ptamd::render and whose 8-bit conversion by
// ptamd::to8Bit (path-tracing_amd/host/reference_binding.hpp).  It is the compile-and-run check
// of that binding against libptamd.so: tests run it on a GPU and compare what it wrote with
// the CPU oracle's render of the flattened scene it also wrote.
//
//   dropin_demo OUT_PREFIX [WIDTH HEIGHT SQRT_SPP]
// writes OUT_PREFIX.img (float RGB, row 0 = bottom), .u8 (to8Bit bytes, same row order),
// .ptscene and .ptcam (what the binding handed to the library).
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#define main reference_main_unused
#include "pathtracer.cpp" // the reference program: every header in its own order; its main is unused
#undef main

#include "pt_amd.h"
#include "reference_binding.hpp"

static bool writeFile(const std::string& name, const void* data, size_t bytes)
{
    FILE* f = fopen(name.c_str(), "wb");
    if (!f)
        return false;
    bool ok = fwrite(data, 1, bytes, f) == bytes;
    fclose(f);
    return ok;
}

int main(int argc, char* argv[])
{
    if (argc < 2) {
        fprintf(stderr, "usage: %s OUT_PREFIX [WIDTH HEIGHT SQRT_SPP [devicebvh]]\n", argv[0]);
        return 2;
    }
    const std::string prefix = argv[1];
    const int width = argc > 2 ? atoi(argv[2]) : 96;
    const int height = argc > 3 ? atoi(argv[3]) : 64;
    const int sqrtSpp = argc > 4 ? atoi(argv[4]) : 3;
    const bool deviceBvh = argc > 5 && std::string(argv[5]) == "devicebvh";
    // `multi`: the image split by tiles over two scene handles (both on GPU 0 when the machine has one)
    const bool multi = argc > 5 && std::string(argv[5]) == "multi";

    Scene scene;
    // a floor: one textured quad under a constant pose
    std::vector<vec3> pos = { vec3(-1, -1, 0), vec3(1, -1, 0), vec3(-1, 1, 0), vec3(1, 1, 0) };
    std::vector<vec3> nrm(4, vec3(0, 0, 1));
    std::vector<vec2> tc = { vec2(0, 0), vec2(1, 0), vec2(0, 1), vec2(1, 1) };
    std::vector<unsigned int> ind = { 0, 1, 2, 1, 3, 2 };
    Texture* checker = scene.take(new TextureChecker(scene.take(new TextureConstant(vec3(0.7f, 0.7f, 0.6f))),
                                                     scene.take(new TextureConstant(vec3(0.2f, 0.25f, 0.3f))), 12, 12));
    Animation* floorPose = scene.take(new AnimationConstant(
        Transformation(vec3(0.0f), quat(radians(-90.0f), vec3(1.0f, 0.0f, 0.0f)), vec3(8.0f))));
    scene.take(new Mesh(pos, nrm, tc, ind, scene.take(new MaterialLambertian(checker)), floorPose));
    // spheres of every kind of material, one of them the light
    scene.take(new SurfaceSphere(vec3(-2.2f, 1.0f, 0.0f), 1.0f,
                                 scene.take(new MaterialMirror(scene.take(new TextureConstant(vec3(0.9f)))))));
    scene.take(new SurfaceSphere(vec3(0.0f, 1.0f, 0.8f), 1.0f, scene.take(new MaterialGlass(vec3(0.2f, 0.05f, 0.05f), 1.5f))));
    scene.take(new SurfaceSphere(vec3(2.2f, 1.0f, 0.0f), 1.0f,
                                 scene.take(new MaterialPhong(scene.take(new TextureConstant(vec3(0.3f, 0.1f, 0.1f))),
                                                              scene.take(new TextureConstant(vec3(0.5f))),
                                                              scene.take(new TextureConstant(vec3(80.0f)))))));
    scene.take(new SurfaceSphere(vec3(0.0f, 0.6f, -2.0f), 0.6f,
                                 scene.take(new MaterialLambertian(scene.take(new TextureConstant(vec3(0.1f, 0.5f, 0.2f)))))));
    scene.take(new SurfaceSphere(vec3(0.0f, 6.0f, 1.0f), 1.2f, scene.take(new MaterialLight(vec3(14.0f, 13.0f, 12.0f)))), true);
    scene.take(new EnvMapEquiRect(scene.take(new TextureConstant(vec3(0.25f, 0.3f, 0.4f)))));
    Animation* camPose = scene.take(new AnimationConstant(Transformation(vec3(0.0f, 3.0f, 9.0f), vec3(0.0f, 1.0f, 0.0f))));
    Camera camera(radians(40.0f), float(width) / height, 9.0f, 0.2f, camPose);

    // (with `devicebvh` the tree is left to ptamd::render, which then builds it on the GPU)
    if (!deviceBvh)
        scene.buildBVH(0.0f, 0.0f);

    // was: the pixel loop of src/15/pathtracer.cpp:199-221
    std::string err;
    std::vector<vec3> img = multi
        ? ptamd::renderMulti(scene, camera, width, height, sqrtSpp, 0.0f, 0.0f, { 0, pt_device_count() > 1 ? 1 : 0 }, &err, 42)
        : ptamd::render(scene, camera, width, height, sqrtSpp, 0.0f, 0.0f, &err, 42);
    if (img.empty()) {
        fprintf(stderr, "render failed: %s\n", err.c_str());
        return 1;
    }
    // was: to8Bit(img), src/15/pathtracer.cpp:225
    std::vector<unsigned char> bytes = ptamd::to8Bit(img, width, height, &err);
    if (bytes.empty()) {
        fprintf(stderr, "to8Bit failed: %s\n", err.c_str());
        return 1;
    }

    // what the binding handed over, for the test to render with the oracle
    if (deviceBvh)
        scene.buildBVH(0.0f, 0.0f);
    ptio::SceneData flat;
    pt_camera cam;
    if (!ptamd::flattenScene(scene, flat, err) || !ptamd::flattenCamera(camera, cam, err)) {
        fprintf(stderr, "flatten failed: %s\n", err.c_str());
        return 1;
    }
    pt_scene_desc desc = flat.desc();
    if (pt_scene_desc_save(&desc, (prefix + ".ptscene").c_str()) != PT_OK) {
        fprintf(stderr, "save failed: %s\n", pt_last_error());
        return 1;
    }
    if (!writeFile(prefix + ".ptcam", &cam, sizeof(cam)) || !writeFile(prefix + ".img", img.data(), img.size() * sizeof(vec3))
        || !writeFile(prefix + ".u8", bytes.data(), bytes.size())) {
        fprintf(stderr, "cannot write output files\n");
        return 1;
    }
    printf("%d x %d, %d spp, %zu surfaces\n", width, height, sqrtSpp * sqrtSpp, scene.surfaces.size());
    return 0;
}
