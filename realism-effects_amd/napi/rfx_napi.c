/*
 * rfx_napi.c — raw N-API addon (node_api.h, no node-gyp / C++ wrappers) that binds the C ABI of
 * librfx_hip.so (include/rfx.h) for the Node host in ../js.
 *
 * Every export is a 1:1 wrapper of one rfx_* entry point; TypedArrays are passed zero-copy
 * (napi_get_typedarray_info).  Pass parameter blocks are plain JS objects whose property names
 * are the reference's uniform / define names (see include/rfx.h for the mapping to the
 * reference files).  Errors become JS exceptions carrying rfx_last_error().
 *
 * All calls are made on the JS main thread; asynchrony comes from the HIP stream inside the
 * context (calls enqueue and return, `sync` blocks), not from JS threads — the same model as the
 * reference, whose draws are asynchronous on the GL command queue.
 */
#include <node_api.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "../../include/rfx.h"

#define NAPI_CALL(env, call)                                          \
    do {                                                              \
        napi_status s__ = (call);                                     \
        if (s__ != napi_ok) {                                         \
            napi_throw_error((env), NULL, "N-API call failed: " #call); \
            return NULL;                                              \
        }                                                             \
    } while (0)

static napi_value throw_rfx(napi_env env, rfx_ctx *c, const char *what, int rc) {
    char buf[640];
    snprintf(buf, sizeof buf, "%s failed (%d): %s", what, rc, rfx_last_error(c));
    napi_throw_error(env, NULL, buf);
    return NULL;
}

static int get_args(napi_env env, napi_callback_info info, size_t want, napi_value *argv) {
    size_t argc = want;
    if (napi_get_cb_info(env, info, &argc, argv, NULL, NULL) != napi_ok || argc < want) {
        napi_throw_type_error(env, NULL, "wrong number of arguments");
        return 0;
    }
    return 1;
}

static rfx_ctx *get_ctx(napi_env env, napi_value v) {
    void *p = NULL;
    if (napi_get_value_external(env, v, &p) != napi_ok || !p) {
        napi_throw_type_error(env, NULL, "expected an rfx context handle");
        return NULL;
    }
    return (rfx_ctx *)p;
}

static int get_int(napi_env env, napi_value v, int32_t *out) {
    napi_valuetype t;
    if (napi_typeof(env, v, &t) != napi_ok) return 0;
    if (t == napi_boolean) { bool b; napi_get_value_bool(env, v, &b); *out = b ? 1 : 0; return 1; }
    if (t != napi_number) return 0;
    double d;
    napi_get_value_double(env, v, &d);
    *out = (int32_t)d;
    return 1;
}

/* obj[name] as double; missing/undefined -> default */
static double prop_num(napi_env env, napi_value obj, const char *name, double def) {
    napi_value v;
    napi_valuetype t;
    if (napi_get_named_property(env, obj, name, &v) != napi_ok || napi_typeof(env, v, &t) != napi_ok) return def;
    if (t == napi_boolean) { bool b; napi_get_value_bool(env, v, &b); return b ? 1.0 : 0.0; }
    if (t != napi_number) return def;
    double d;
    napi_get_value_double(env, v, &d);
    return d;
}

/* obj[name] = Float32Array | Float64Array | Array of n numbers -> out[n] */
static int prop_floats(napi_env env, napi_value obj, const char *name, float *out, size_t n) {
    napi_value v;
    bool is_ta = false, is_arr = false;
    if (napi_get_named_property(env, obj, name, &v) != napi_ok) return 0;
    napi_is_typedarray(env, v, &is_ta);
    if (is_ta) {
        napi_typedarray_type ty; size_t len; void *data;
        if (napi_get_typedarray_info(env, v, &ty, &len, &data, NULL, NULL) != napi_ok || len < n) return 0;
        if (ty == napi_float32_array) { memcpy(out, data, n * sizeof(float)); return 1; }
        if (ty == napi_float64_array) { for (size_t i = 0; i < n; i++) out[i] = (float)((double *)data)[i]; return 1; }
        return 0;
    }
    napi_is_array(env, v, &is_arr);
    if (!is_arr) return 0;
    for (size_t i = 0; i < n; i++) {
        napi_value e; double d;
        if (napi_get_element(env, v, (uint32_t)i, &e) != napi_ok || napi_get_value_double(env, e, &d) != napi_ok) return 0;
        out[i] = (float)d;
    }
    return 1;
}

static int prop_bool2(napi_env env, napi_value obj, const char *name, int32_t *out) {
    napi_value v, e;
    bool is_arr = false;
    out[0] = out[1] = 0;
    if (napi_get_named_property(env, obj, name, &v) != napi_ok) return 0;
    napi_is_array(env, v, &is_arr);
    if (!is_arr) { int32_t x = 0; if (get_int(env, v, &x)) { out[0] = out[1] = x; return 1; } return 0; }
    for (uint32_t i = 0; i < 2; i++)
        if (napi_get_element(env, v, i, &e) == napi_ok) get_int(env, e, &out[i]);
    return 1;
}

/* camera object: the fields the reference passes read from three's camera */
static int read_camera(napi_env env, napi_value obj, const char *name, rfx_camera *c) {
    napi_value cam;
    napi_valuetype t;
    if (napi_get_named_property(env, obj, name, &cam) != napi_ok || napi_typeof(env, cam, &t) != napi_ok || t != napi_object) {
        napi_throw_type_error(env, NULL, "params.camera / params.prevCamera must be an object");
        return 0;
    }
    if (!prop_floats(env, cam, "projectionMatrix", c->projectionMatrix, 16) || !prop_floats(env, cam, "projectionMatrixInverse", c->projectionMatrixInverse, 16) ||
        !prop_floats(env, cam, "matrixWorld", c->matrixWorld, 16) || !prop_floats(env, cam, "matrixWorldInverse", c->matrixWorldInverse, 16) ||
        !prop_floats(env, cam, "position", c->position, 3)) {
        napi_throw_type_error(env, NULL, "camera needs projectionMatrix, projectionMatrixInverse, matrixWorld, matrixWorldInverse (16 numbers) and position (3)");
        return 0;
    }
    c->near_ = (float)prop_num(env, cam, "near", 0.1);
    c->far_ = (float)prop_num(env, cam, "far", 1000.0);
    c->isPerspective = (int32_t)prop_num(env, cam, "isPerspectiveCamera", 1.0);
    return 1;
}

static void destroy_ctx(napi_env env, void *data, void *hint) { (void)env; (void)hint; rfx_destroy((rfx_ctx *)data); }

/* create(device, width, height, tileY0, tileRows, haloRows) -> handle */
static napi_value n_create(napi_env env, napi_callback_info info) {
    napi_value a[6], out;
    int32_t v[6];
    if (!get_args(env, info, 6, a)) return NULL;
    for (int i = 0; i < 6; i++) if (!get_int(env, a[i], &v[i])) { napi_throw_type_error(env, NULL, "create: integers expected"); return NULL; }
    rfx_ctx *c = rfx_create(v[0], v[1], v[2], v[3], v[4], v[5]);
    if (!c) return throw_rfx(env, NULL, "rfx_create", RFX_EDEVICE);
    NAPI_CALL(env, napi_create_external(env, c, destroy_ctx, NULL, &out));
    return out;
}

static napi_value n_abi_version(napi_env env, napi_callback_info info) {
    (void)info;
    napi_value out;
    NAPI_CALL(env, napi_create_int32(env, rfx_abi_version(), &out));
    return out;
}

/* heldRows(ctx, tex) -> [row0, rows] */
static napi_value n_held_rows(napi_env env, napi_callback_info info) {
    napi_value a[2], arr, e;
    int32_t tex;
    if (!get_args(env, info, 2, a)) return NULL;
    rfx_ctx *c = get_ctx(env, a[0]);
    if (!c || !get_int(env, a[1], &tex)) return NULL;
    int r0 = 0, n = 0;
    int rc = rfx_tex_held_rows(c, (rfx_tex)tex, &r0, &n);
    if (rc) return throw_rfx(env, c, "rfx_tex_held_rows", rc);
    NAPI_CALL(env, napi_create_array_with_length(env, 2, &arr));
    napi_create_int32(env, r0, &e); napi_set_element(env, arr, 0, e);
    napi_create_int32(env, n, &e); napi_set_element(env, arr, 1, e);
    return arr;
}

/* upload(ctx, tex, typedArray, row0, rows) / download(ctx, tex, typedArray, row0, rows) */
static napi_value xfer(napi_env env, napi_callback_info info, int up) {
    napi_value a[5];
    int32_t tex, row0, rows;
    if (!get_args(env, info, 5, a)) return NULL;
    rfx_ctx *c = get_ctx(env, a[0]);
    if (!c || !get_int(env, a[1], &tex) || !get_int(env, a[3], &row0) || !get_int(env, a[4], &rows)) return NULL;
    bool is_ta = false;
    napi_is_typedarray(env, a[2], &is_ta);
    if (!is_ta) { napi_throw_type_error(env, NULL, "upload/download: TypedArray expected"); return NULL; }
    napi_typedarray_type ty; size_t len; void *data; napi_value ab; size_t off;
    NAPI_CALL(env, napi_get_typedarray_info(env, a[2], &ty, &len, &data, &ab, &off));
    static const size_t esz[] = {1, 1, 1, 2, 2, 4, 4, 4, 8, 8, 8};
    const size_t bytes = len * esz[ty];
    int width = 0;
    rfx_get_geometry(c, &width, NULL, NULL, NULL, NULL);
    if (tex == RFX_TEX_BLUE_NOISE) width = 128;
    const size_t need = (size_t)rows * (size_t)width * rfx_tex_texel_bytes((rfx_tex)tex);
    if (rows <= 0 || need == 0 || bytes != need) {
        napi_throw_range_error(env, NULL, "upload/download: TypedArray byte length != rows * width * texel bytes");
        return NULL;
    }
    int rc = up == 2 ? rfx_stage_upload(c, (rfx_tex)tex, data, row0, rows)
             : up ? rfx_upload(c, (rfx_tex)tex, data, row0, rows) : rfx_download(c, (rfx_tex)tex, data, row0, rows);
    if (rc) return throw_rfx(env, c, up == 2 ? "rfx_stage_upload" : up ? "rfx_upload" : "rfx_download", rc);
    return NULL;
}
/* stageUpload(ctx, tex, typedArray, row0, rows): asynchronous copy into the slot's back buffer (rfx.h "streaming dumps"); the array —
 * ideally a view of hostAlloc() memory — must stay alive and unchanged until the flip that publishes it has been synced */
static napi_value n_stage_upload(napi_env env, napi_callback_info info) { return xfer(env, info, 2); }
static napi_value n_stage_flip(napi_env env, napi_callback_info info) {
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return NULL;
    rfx_ctx *c = get_ctx(env, a[0]);
    if (!c) return NULL;
    int rc = rfx_stage_flip(c);
    if (rc) return throw_rfx(env, c, "rfx_stage_flip", rc);
    return NULL;
}
static void host_free_cb(napi_env env, void *data, void *hint) { (void)env; (void)hint; rfx_host_free(data); }
/* hostAlloc(bytes) -> ArrayBuffer over pinned host memory (hipHostMalloc), freed when the buffer is collected */
static napi_value n_host_alloc(napi_env env, napi_callback_info info) {
    napi_value a[1], ab;
    double bytes = 0;
    if (!get_args(env, info, 1, a) || napi_get_value_double(env, a[0], &bytes) != napi_ok || bytes <= 0) {
        napi_throw_range_error(env, NULL, "hostAlloc: a positive byte count");
        return NULL;
    }
    void *p = rfx_host_alloc((size_t)bytes);
    if (!p) { napi_throw_error(env, NULL, "rfx_host_alloc failed"); return NULL; }
    if (napi_create_external_arraybuffer(env, p, (size_t)bytes, host_free_cb, NULL, &ab) != napi_ok) {
        rfx_host_free(p);
        napi_throw_error(env, NULL, "hostAlloc: napi_create_external_arraybuffer failed");
        return NULL;
    }
    return ab;
}
static napi_value n_upload(napi_env env, napi_callback_info info) { return xfer(env, info, 1); }
static napi_value n_download(napi_env env, napi_callback_info info) { return xfer(env, info, 0); }

static napi_value n_clear(napi_env env, napi_callback_info info) {
    napi_value a[2];
    int32_t tex;
    if (!get_args(env, info, 2, a)) return NULL;
    rfx_ctx *c = get_ctx(env, a[0]);
    if (!c || !get_int(env, a[1], &tex)) return NULL;
    int rc = rfx_clear(c, (rfx_tex)tex);
    if (rc) return throw_rfx(env, c, "rfx_clear", rc);
    return NULL;
}

static const float *f32_prop(napi_env env, napi_value obj, const char *name, size_t want, int optional, int *ok) {
    napi_value v;
    bool has = false;
    napi_valuetype vt = napi_undefined;
    if (napi_has_named_property(env, obj, name, &has) == napi_ok && has && napi_get_named_property(env, obj, name, &v) == napi_ok) napi_typeof(env, v, &vt);
    if (!has || vt == napi_null || vt == napi_undefined) {
        if (!optional) { napi_throw_type_error(env, NULL, "AOV plane missing"); *ok = 0; }
        return NULL;
    }
    napi_typedarray_type tt;
    size_t len;
    void *ptr;
    if (napi_get_typedarray_info(env, v, &tt, &len, &ptr, NULL, NULL) != napi_ok || tt != napi_float32_array || len != want) {
        napi_throw_type_error(env, NULL, "AOV plane: Float32Array of rows * width * channels expected");
        *ok = 0;
        return NULL;
    }
    return (const float *)ptr;
}

/* packGBuffer(ctx, {diffuse, normal, roughness, metalness, emissive, depth?}, row0, rows) / packVelocity(ctx, {velocity, normal, depth}, row0, rows) */
static napi_value n_pack(napi_env env, napi_callback_info info, int velocity) {
    napi_value a[4];
    int32_t row0, rows, W = 0;
    if (!get_args(env, info, 4, a)) return NULL;
    rfx_ctx *c = get_ctx(env, a[0]);
    if (!c || !get_int(env, a[2], &row0) || !get_int(env, a[3], &rows)) return NULL;
    rfx_get_geometry(c, &W, NULL, NULL, NULL, NULL);
    const size_t n = (size_t)rows * (size_t)W;
    int ok = 1, rc;
    if (velocity) {
        rfx_aov_velocity v = {f32_prop(env, a[1], "velocity", 2 * n, 0, &ok), f32_prop(env, a[1], "normal", 3 * n, 0, &ok), f32_prop(env, a[1], "depth", n, 0, &ok)};
        if (!ok) return NULL;
        rc = rfx_pack_velocity(c, &v, row0, rows);
    } else {
        rfx_aov_gbuffer g = {f32_prop(env, a[1], "diffuse", 4 * n, 0, &ok), f32_prop(env, a[1], "normal", 3 * n, 0, &ok), f32_prop(env, a[1], "roughness", n, 0, &ok),
                             f32_prop(env, a[1], "metalness", n, 0, &ok), f32_prop(env, a[1], "emissive", 3 * n, 0, &ok), f32_prop(env, a[1], "depth", n, 1, &ok)};
        if (!ok) return NULL;
        rc = rfx_pack_gbuffer(c, &g, row0, rows);
    }
    if (rc) return throw_rfx(env, c, velocity ? "rfx_pack_velocity" : "rfx_pack_gbuffer", rc);
    return NULL;
}
static napi_value n_pack_gbuffer(napi_env env, napi_callback_info info) { return n_pack(env, info, 0); }
static napi_value n_pack_velocity(napi_env env, napi_callback_info info) { return n_pack(env, info, 1); }

/* setEnvironment(ctx, Float32Array | null, width, height, halfFloatType, halfStoreRTZ) — scene.environment (rfx_set_environment) */
static napi_value n_set_environment(napi_env env, napi_callback_info info) {
    napi_value a[6];
    if (!get_args(env, info, 6, a)) return NULL;
    rfx_ctx *c = get_ctx(env, a[0]);
    if (!c) return NULL;
    napi_valuetype vt;
    napi_typeof(env, a[1], &vt);
    int32_t w = 0, h = 0, half = 1, rtz = 1;
    const float *data = NULL;
    if (vt != napi_null && vt != napi_undefined) {
        napi_typedarray_type tt;
        size_t len;
        void *ptr;
        if (napi_get_typedarray_info(env, a[1], &tt, &len, &ptr, NULL, NULL) != napi_ok || tt != napi_float32_array) {
            napi_throw_type_error(env, NULL, "setEnvironment: Float32Array or null expected");
            return NULL;
        }
        if (!get_int(env, a[2], &w) || !get_int(env, a[3], &h) || !get_int(env, a[4], &half) || !get_int(env, a[5], &rtz)) return NULL;
        if ((size_t)w * (size_t)h * 4 != len) {
            napi_throw_range_error(env, NULL, "setEnvironment: data length != width * height * 4");
            return NULL;
        }
        data = (const float *)ptr;
    }
    int rc = rfx_set_environment(c, data, w, h, half, rtz);
    if (rc) return throw_rfx(env, c, "rfx_set_environment", rc);
    return NULL;
}

/* cubeToEquirect(ctx, Float32Array faces[6 * size * size * 4], size, generateMipmaps, Float32Array out[width * height * 4], width, height):
 * rfx_cube_to_equirect (CubeToEquirectEnvPass's draw + read-back) */
static napi_value n_cube_to_equirect(napi_env env, napi_callback_info info) {
    napi_value a[7];
    if (!get_args(env, info, 7, a)) return NULL;
    rfx_ctx *c = get_ctx(env, a[0]);
    if (!c) return NULL;
    int32_t size, mips, w, h;
    if (!get_int(env, a[2], &size) || !get_int(env, a[3], &mips) || !get_int(env, a[5], &w) || !get_int(env, a[6], &h)) return NULL;
    napi_typedarray_type tt;
    size_t len_in, len_out;
    void *pin, *pout;
    if (napi_get_typedarray_info(env, a[1], &tt, &len_in, &pin, NULL, NULL) != napi_ok || tt != napi_float32_array ||
        napi_get_typedarray_info(env, a[4], &tt, &len_out, &pout, NULL, NULL) != napi_ok || tt != napi_float32_array) {
        napi_throw_type_error(env, NULL, "cubeToEquirect: Float32Array faces and Float32Array target expected");
        return NULL;
    }
    if (size < 1 || w < 1 || h < 1 || (size_t)6 * (size_t)size * (size_t)size * 4 != len_in || (size_t)w * (size_t)h * 4 != len_out) {
        napi_throw_range_error(env, NULL, "cubeToEquirect: faces length != 6 * size * size * 4 or target length != width * height * 4");
        return NULL;
    }
    int rc = rfx_cube_to_equirect(c, (const float *)pin, size, mips, (float *)pout, w, h);
    if (rc) return throw_rfx(env, c, "rfx_cube_to_equirect", rc);
    return NULL;
}

/* setEnvironmentImportance(ctx, Float32Array marginal, Float32Array conditional, totalSumWhole, totalSumDecimal) */
static napi_value n_set_environment_importance(napi_env env, napi_callback_info info) {
    napi_value a[5];
    if (!get_args(env, info, 5, a)) return NULL;
    rfx_ctx *c = get_ctx(env, a[0]);
    if (!c) return NULL;
    const float *tab[2];
    size_t count[2];
    for (int k = 0; k < 2; k++) {
        napi_typedarray_type tt;
        size_t len;
        void *ptr;
        if (napi_get_typedarray_info(env, a[1 + k], &tt, &len, &ptr, NULL, NULL) != napi_ok || tt != napi_float32_array) {
            napi_throw_type_error(env, NULL, "setEnvironmentImportance: Float32Array expected");
            return NULL;
        }
        tab[k] = (const float *)ptr;
        count[k] = len;  /* the library checks them against the environment's size before it copies */
    }
    double whole = 0, dec = 0;
    if (napi_get_value_double(env, a[3], &whole) != napi_ok || napi_get_value_double(env, a[4], &dec) != napi_ok) {
        napi_throw_type_error(env, NULL, "setEnvironmentImportance: totalSumWhole / totalSumDecimal must be numbers");
        return NULL;
    }
    int rc = rfx_set_environment_importance(c, tab[0], count[0], tab[1], count[1], (float)whole, (float)dec);
    if (rc) return throw_rfx(env, c, "rfx_set_environment_importance", rc);
    return NULL;
}

/* ssgiMarch / ssgiTrace / ssgiShade(ctx, {camera, steps, refineSteps, mode, useDirectLight, missedRays, importanceSampling,
 *                 rayDistance, thickness, envBlur, blueNoiseIndex, historySource, resolutionScale, useEnvMap}) */
static napi_value ssgi_stage(napi_env env, napi_callback_info info, int stage) {
    napi_value a[2];
    if (!get_args(env, info, 2, a)) return NULL;
    rfx_ctx *c = get_ctx(env, a[0]);
    if (!c) return NULL;
    rfx_ssgi_params p;
    memset(&p, 0, sizeof p);
    if (!read_camera(env, a[1], "camera", &p.camera)) return NULL;
    p.steps = (int32_t)prop_num(env, a[1], "steps", 20);
    p.refineSteps = (int32_t)prop_num(env, a[1], "refineSteps", 5);
    p.mode = (int32_t)prop_num(env, a[1], "mode", 0);
    p.useDirectLight = (int32_t)prop_num(env, a[1], "useDirectLight", 0);
    p.missedRays = (int32_t)prop_num(env, a[1], "missedRays", 0);
    p.importanceSampling = (int32_t)prop_num(env, a[1], "importanceSampling", 0);
    p.useEnvMap = (int32_t)prop_num(env, a[1], "useEnvMap", 0);
    p.rayDistance = (float)prop_num(env, a[1], "rayDistance", 10);
    p.thickness = (float)prop_num(env, a[1], "thickness", 10);
    p.envBlur = (float)prop_num(env, a[1], "envBlur", 0.5);
    p.blueNoiseIndex = (int32_t)prop_num(env, a[1], "blueNoiseIndex", 0);
    p.historySource = (int32_t)prop_num(env, a[1], "historySource", 0);
    p.resolutionScale = (float)prop_num(env, a[1], "resolutionScale", 1);
    int rc = stage == 0 ? rfx_ssgi_march(c, &p) : (stage == 1 ? rfx_ssgi_trace(c, &p) : rfx_ssgi_shade(c, &p));
    if (rc) return throw_rfx(env, c, stage == 0 ? "rfx_ssgi_march" : (stage == 1 ? "rfx_ssgi_trace" : "rfx_ssgi_shade"), rc);
    return NULL;
}
static napi_value n_ssgi(napi_env env, napi_callback_info info) { return ssgi_stage(env, info, 0); }
static napi_value n_ssgi_trace(napi_env env, napi_callback_info info) { return ssgi_stage(env, info, 1); }
static napi_value n_ssgi_shade(napi_env env, napi_callback_info info) { return ssgi_stage(env, info, 2); }

/* temporalReproject(ctx, {camera, prevCamera, textureCount, inputType, reprojectSpecular[2], neighborhoodClamp[2],
 *                         logTransform, fullAccumulate, confidencePower, neighborhoodClampIntensity, maxBlend, keepData,
 *                         historySource, targetHalf, halfStoreRTZ}) */
static napi_value n_temporal(napi_env env, napi_callback_info info) {
    napi_value a[2];
    if (!get_args(env, info, 2, a)) return NULL;
    rfx_ctx *c = get_ctx(env, a[0]);
    if (!c) return NULL;
    rfx_temporal_params p;
    memset(&p, 0, sizeof p);
    if (!read_camera(env, a[1], "camera", &p.camera) || !read_camera(env, a[1], "prevCamera", &p.prevCamera)) return NULL;
    p.textureCount = (int32_t)prop_num(env, a[1], "textureCount", 2);
    p.inputType = (int32_t)prop_num(env, a[1], "inputType", 0);
    prop_bool2(env, a[1], "reprojectSpecular", p.reprojectSpecular);
    prop_bool2(env, a[1], "neighborhoodClamp", p.neighborhoodClamp);
    p.logTransform = (int32_t)prop_num(env, a[1], "logTransform", 0);
    p.fullAccumulate = (int32_t)prop_num(env, a[1], "fullAccumulate", 0);
    p.confidencePower = (float)prop_num(env, a[1], "confidencePower", 0.75);
    p.neighborhoodClampIntensity = (float)prop_num(env, a[1], "neighborhoodClampIntensity", 1);
    p.maxBlend = (float)prop_num(env, a[1], "maxBlend", 1);
    p.keepData = (float)prop_num(env, a[1], "keepData", 1);
    p.historySource = (int32_t)prop_num(env, a[1], "historySource", 0);
    p.targetHalf = (int32_t)prop_num(env, a[1], "targetHalf", 0);
    p.halfStoreRTZ = (int32_t)prop_num(env, a[1], "halfStoreRTZ", 1);
    p.inputWidth = (int32_t)prop_num(env, a[1], "inputWidth", 0);
    p.inputHeight = (int32_t)prop_num(env, a[1], "inputHeight", 0);
    int rc = rfx_temporal_reproject(c, &p);
    if (rc) return throw_rfx(env, c, "rfx_temporal_reproject", rc);
    return NULL;
}

/* copyFramebuffer(ctx, dstTex) — renderer.copyFramebufferToTexture of TemporalReprojectPass.js:198-201 */
/* setRowWindow(ctx, y0, y1): rfx_set_row_window (y1 <= y0 resets) */
static napi_value n_set_row_window(napi_env env, napi_callback_info info) {
    napi_value a[3];
    int32_t y0, y1;
    if (!get_args(env, info, 3, a)) return NULL;
    rfx_ctx *c = get_ctx(env, a[0]);
    if (!c || !get_int(env, a[1], &y0) || !get_int(env, a[2], &y1)) return NULL;
    int rc = rfx_set_row_window(c, y0, y1);
    if (rc) return throw_rfx(env, c, "rfx_set_row_window", rc);
    return NULL;
}

/* setUvModel(ctx, model): rfx_set_uv_model (0 = RFX_UV_IDEAL, 1 = RFX_UV_REFERENCE_GL) */
static napi_value n_set_uv_model(napi_env env, napi_callback_info info) {
    napi_value a[2];
    int32_t m;
    if (!get_args(env, info, 2, a)) return NULL;
    rfx_ctx *c = get_ctx(env, a[0]);
    if (!c || !get_int(env, a[1], &m)) return NULL;
    int rc = rfx_set_uv_model(c, m);
    if (rc) return throw_rfx(env, c, "rfx_set_uv_model", rc);
    return NULL;
}

/* ---- row-tiled runs (rfx.h "row-tiled runs"): one Node process per GPU */
/* splitRows(height, nranks, rank) -> [tile_y0, tile_rows] */
static napi_value n_split_rows(napi_env env, napi_callback_info info) {
    napi_value a[3], arr, v;
    int32_t h, n, r;
    int y0 = 0, rows = 0;
    if (!get_args(env, info, 3, a) || !get_int(env, a[0], &h) || !get_int(env, a[1], &n) || !get_int(env, a[2], &r)) return NULL;
    if (rfx_split_rows(h, n, r, &y0, &rows) != RFX_OK) {
        napi_throw_range_error(env, NULL, "splitRows: height cannot be cut into that many tiles of at least 2 rows");
        return NULL;
    }
    NAPI_CALL(env, napi_create_array_with_length(env, 2, &arr));
    NAPI_CALL(env, napi_create_int32(env, y0, &v));
    napi_set_element(env, arr, 0, v);
    NAPI_CALL(env, napi_create_int32(env, rows, &v));
    napi_set_element(env, arr, 1, v);
    return arr;
}
/* commUniqueId() -> Buffer(128) (ncclGetUniqueId; rank 0 hands it to the other processes, e.g. through a file) */
static napi_value n_comm_unique_id(napi_env env, napi_callback_info info) {
    char id[128];
    napi_value buf;
    void *data = NULL;
    int rc = rfx_comm_unique_id(id);
    if (rc) {
        napi_throw_error(env, NULL, "rfx_comm_unique_id failed: RCCL not loadable on this host or no device");
        return NULL;
    }
    NAPI_CALL(env, napi_create_buffer_copy(env, sizeof id, id, &data, &buf));
    return buf;
}
/* commInit(ctx, Buffer id128, rank, nranks) */
static napi_value n_comm_init(napi_env env, napi_callback_info info) {
    napi_value a[4];
    int32_t rank, n;
    void *data = NULL;
    size_t len = 0;
    if (!get_args(env, info, 4, a)) return NULL;
    rfx_ctx *c = get_ctx(env, a[0]);
    if (!c || !get_int(env, a[2], &rank) || !get_int(env, a[3], &n)) return NULL;
    if (napi_get_buffer_info(env, a[1], &data, &len) != napi_ok || len != 128) {
        napi_throw_type_error(env, NULL, "commInit: the unique id is a 128-byte Buffer");
        return NULL;
    }
    int rc = rfx_comm_init(c, data, rank, n);
    if (rc) return throw_rfx(env, c, "rfx_comm_init", rc);
    return NULL;
}
/* haloExchange(ctx, tex, upRank, downRank) — -1 = no such neighbour */
static napi_value n_halo_exchange(napi_env env, napi_callback_info info) {
    napi_value a[4];
    int32_t tex, up, down;
    if (!get_args(env, info, 4, a)) return NULL;
    rfx_ctx *c = get_ctx(env, a[0]);
    if (!c || !get_int(env, a[1], &tex) || !get_int(env, a[2], &up) || !get_int(env, a[3], &down)) return NULL;
    int rc = rfx_halo_exchange(c, (rfx_tex)tex, NULL, up, down);
    if (rc) return throw_rfx(env, c, "rfx_halo_exchange", rc);
    return NULL;
}
/* allgatherHistory(ctx, tex) */
static napi_value n_allgather_history(napi_env env, napi_callback_info info) {
    napi_value a[2];
    int32_t tex;
    if (!get_args(env, info, 2, a)) return NULL;
    rfx_ctx *c = get_ctx(env, a[0]);
    if (!c || !get_int(env, a[1], &tex)) return NULL;
    int rc = rfx_allgather_history(c, (rfx_tex)tex, NULL);
    if (rc) return throw_rfx(env, c, "rfx_allgather_history", rc);
    return NULL;
}
/* gatherHistoryRows(ctx, tex) -> bytes this rank receives (rfx_gather_history_rows: between ssgiTrace and ssgiShade) */
static napi_value n_gather_history_rows(napi_env env, napi_callback_info info) {
    napi_value a[2], out;
    int32_t tex;
    size_t got = 0;
    if (!get_args(env, info, 2, a)) return NULL;
    rfx_ctx *c = get_ctx(env, a[0]);
    if (!c || !get_int(env, a[1], &tex)) return NULL;
    int rc = rfx_gather_history_rows(c, (rfx_tex)tex, NULL, &got);
    if (rc) return throw_rfx(env, c, "rfx_gather_history_rows", rc);
    napi_create_double(env, (double)got, &out);
    return out;
}
/* The device-driven history gather (include/rfx.h rfx_peer_*: the consumer's own kernel loads the column blocks its rays read out of the
 * peers' planes through HIP IPC mappings).  peerExport(ctx, tex) -> Buffer of RFX_PEER_BLOB_BYTES; peerOpen(ctx, tex, Buffer of nranks blobs in
 * rank order, rank, nranks); peerGatherHistory(ctx, tex) -> bytes the PREVIOUS call pulled; peerClose(ctx) */
static napi_value n_peer_export(napi_env env, napi_callback_info info) {
    napi_value a[2], buf;
    int32_t tex;
    char blob[RFX_PEER_BLOB_BYTES];
    void *data = NULL;
    if (!get_args(env, info, 2, a)) return NULL;
    rfx_ctx *c = get_ctx(env, a[0]);
    if (!c || !get_int(env, a[1], &tex)) return NULL;
    int rc = rfx_peer_export(c, (rfx_tex)tex, blob);
    if (rc) return throw_rfx(env, c, "rfx_peer_export", rc);
    NAPI_CALL(env, napi_create_buffer_copy(env, sizeof blob, blob, &data, &buf));
    return buf;
}
static napi_value n_peer_open(napi_env env, napi_callback_info info) {
    napi_value a[5];
    int32_t tex, rank, n;
    void *data = NULL;
    size_t len = 0;
    if (!get_args(env, info, 5, a)) return NULL;
    rfx_ctx *c = get_ctx(env, a[0]);
    if (!c || !get_int(env, a[1], &tex) || !get_int(env, a[3], &rank) || !get_int(env, a[4], &n)) return NULL;
    if (n < 1 || napi_get_buffer_info(env, a[2], &data, &len) != napi_ok || len != (size_t)n * RFX_PEER_BLOB_BYTES) {
        napi_throw_type_error(env, NULL, "peerOpen: the blobs are one Buffer of nranks * RFX_PEER_BLOB_BYTES bytes, in rank order");
        return NULL;
    }
    int rc = rfx_peer_open(c, (rfx_tex)tex, data, rank, n);
    if (rc) return throw_rfx(env, c, "rfx_peer_open", rc);
    return NULL;
}
static napi_value n_peer_gather_history(napi_env env, napi_callback_info info) {
    napi_value a[2], out;
    int32_t tex;
    size_t got = 0;
    if (!get_args(env, info, 2, a)) return NULL;
    rfx_ctx *c = get_ctx(env, a[0]);
    if (!c || !get_int(env, a[1], &tex)) return NULL;
    int rc = rfx_peer_gather_history(c, (rfx_tex)tex, &got);
    if (rc) return throw_rfx(env, c, "rfx_peer_gather_history", rc);
    napi_create_double(env, (double)got, &out);
    return out;
}
static napi_value n_peer_close(napi_env env, napi_callback_info info) {
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return NULL;
    rfx_ctx *c = get_ctx(env, a[0]);
    if (!c) return NULL;
    int rc = rfx_peer_close(c);
    if (rc) return throw_rfx(env, c, "rfx_peer_close", rc);
    return NULL;
}
/* ssgiHitMask(ctx, Uint32Array of frame-height entries): rfx_ssgi_hit_mask (after ssgiTrace; blocks until the trace has finished) */
static napi_value n_ssgi_hit_mask(napi_env env, napi_callback_info info) {
    napi_value a[2], ab;
    napi_typedarray_type type;
    size_t len = 0, off = 0;
    void *data = NULL;
    if (!get_args(env, info, 2, a)) return NULL;
    rfx_ctx *c = get_ctx(env, a[0]);
    if (!c) return NULL;
    if (napi_get_typedarray_info(env, a[1], &type, &len, &data, &ab, &off) != napi_ok || type != napi_uint32_array) {
        napi_throw_type_error(env, NULL, "ssgiHitMask: a Uint32Array with one entry per frame row");
        return NULL;
    }
    int rc = rfx_ssgi_hit_mask(c, (unsigned int *)data, (int)len);
    if (rc) return throw_rfx(env, c, "rfx_ssgi_hit_mask", rc);
    return NULL;
}
/* commWait(ctx) / commDestroy(ctx) */
static napi_value n_comm_wait(napi_env env, napi_callback_info info) {
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return NULL;
    rfx_ctx *c = get_ctx(env, a[0]);
    if (!c) return NULL;
    int rc = rfx_comm_wait(c);
    if (rc) return throw_rfx(env, c, "rfx_comm_wait", rc);
    return NULL;
}
static napi_value n_comm_destroy(napi_env env, napi_callback_info info) {
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return NULL;
    rfx_ctx *c = get_ctx(env, a[0]);
    if (!c) return NULL;
    rfx_comm_destroy(c);
    return NULL;
}

static napi_value n_copy_framebuffer(napi_env env, napi_callback_info info) {
    napi_value a[2];
    int32_t tex;
    if (!get_args(env, info, 2, a)) return NULL;
    rfx_ctx *c = get_ctx(env, a[0]);
    if (!c || !get_int(env, a[1], &tex)) return NULL;
    int rc = rfx_copy_framebuffer(c, (rfx_tex)tex);
    if (rc) return throw_rfx(env, c, "rfx_copy_framebuffer", rc);
    return NULL;
}

/* poissonDenoise(ctx, {radius, phi, lumaPhi, depthPhi, normalPhi, roughnessPhi, specularPhi, textureCount,
 *                      isTextureSpecular[2], blueNoiseIndex, inputIsTemporal, writeToB, halfStoreRTZ}) */
static napi_value n_denoise(napi_env env, napi_callback_info info) {
    napi_value a[2];
    if (!get_args(env, info, 2, a)) return NULL;
    rfx_ctx *c = get_ctx(env, a[0]);
    if (!c) return NULL;
    rfx_denoise_params p;
    memset(&p, 0, sizeof p);
    p.radius = (float)prop_num(env, a[1], "radius", 3);
    p.phi = (float)prop_num(env, a[1], "phi", 0.5);
    p.lumaPhi = (float)prop_num(env, a[1], "lumaPhi", 5);
    p.depthPhi = (float)prop_num(env, a[1], "depthPhi", 2);
    p.normalPhi = (float)prop_num(env, a[1], "normalPhi", 3.25);
    p.roughnessPhi = (float)prop_num(env, a[1], "roughnessPhi", 0.0 / 0.0);
    p.specularPhi = (float)prop_num(env, a[1], "specularPhi", 0.0 / 0.0);
    p.textureCount = (int32_t)prop_num(env, a[1], "textureCount", 2);
    prop_bool2(env, a[1], "isTextureSpecular", p.isTextureSpecular);
    p.blueNoiseIndex = (int32_t)prop_num(env, a[1], "blueNoiseIndex", 0);
    p.inputIsTemporal = (int32_t)prop_num(env, a[1], "inputIsTemporal", 1);
    p.writeToB = (int32_t)prop_num(env, a[1], "writeToB", 0);
    p.halfStoreRTZ = (int32_t)prop_num(env, a[1], "halfStoreRTZ", 0);
    int rc = rfx_poisson_denoise(c, &p);
    if (rc) return throw_rfx(env, c, "rfx_poisson_denoise", rc);
    return NULL;
}

/* compose(ctx, {camera, inputType}) */
static napi_value n_compose(napi_env env, napi_callback_info info) {
    napi_value a[2];
    if (!get_args(env, info, 2, a)) return NULL;
    rfx_ctx *c = get_ctx(env, a[0]);
    if (!c) return NULL;
    rfx_compose_params p;
    memset(&p, 0, sizeof p);
    if (!read_camera(env, a[1], "camera", &p.camera)) return NULL;
    p.inputType = (int32_t)prop_num(env, a[1], "inputType", 0);
    p.giSource = (int32_t)prop_num(env, a[1], "giSource", 0);
    p.writeHistoryRGB = (int32_t)prop_num(env, a[1], "writeHistoryRGB", 0);
    int rc = rfx_compose(c, &p);
    if (rc) return throw_rfx(env, c, "rfx_compose", rc);
    return NULL;
}

/* finalCompose(ctx, {camera, isDebug, inputSource, fogMode, fogColor[3], fogNear, fogFar, fogDensity}) — SSGIEffect's own fragment */
static napi_value n_final(napi_env env, napi_callback_info info) {
    napi_value a[2];
    if (!get_args(env, info, 2, a)) return NULL;
    rfx_ctx *c = get_ctx(env, a[0]);
    if (!c) return NULL;
    rfx_final_params p;
    memset(&p, 0, sizeof p);
    if (!read_camera(env, a[1], "camera", &p.camera)) return NULL;
    p.isDebug = (int32_t)prop_num(env, a[1], "isDebug", 0);
    p.inputSource = (int32_t)prop_num(env, a[1], "inputSource", 0);
    p.fogMode = (int32_t)prop_num(env, a[1], "fogMode", 0);
    napi_value fc;
    bool has = false;
    if (napi_has_named_property(env, a[1], "fogColor", &has) == napi_ok && has && napi_get_named_property(env, a[1], "fogColor", &fc) == napi_ok) {
        for (uint32_t k = 0; k < 3; k++) {
            napi_value e;
            double v = 0;
            if (napi_get_element(env, fc, k, &e) == napi_ok) napi_get_value_double(env, e, &v);
            p.fogColor[k] = (float)v;
        }
    }
    p.fogNear = (float)prop_num(env, a[1], "fogNear", 0);
    p.fogFar = (float)prop_num(env, a[1], "fogFar", 0);
    p.fogDensity = (float)prop_num(env, a[1], "fogDensity", 0);
    int rc = rfx_final_compose(c, &p);
    if (rc) return throw_rfx(env, c, "rfx_final_compose", rc);
    return NULL;
}

static napi_value n_sync(napi_env env, napi_callback_info info) {
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return NULL;
    rfx_ctx *c = get_ctx(env, a[0]);
    if (!c) return NULL;
    int rc = rfx_sync(c);
    if (rc) return throw_rfx(env, c, "rfx_sync", rc);
    return NULL;
}

static napi_value n_halo_violations(napi_env env, napi_callback_info info) {
    napi_value a[1], out;
    if (!get_args(env, info, 1, a)) return NULL;
    rfx_ctx *c = get_ctx(env, a[0]);
    if (!c) return NULL;
    NAPI_CALL(env, napi_create_uint32(env, rfx_halo_violations(c), &out));
    return out;
}

/* timeBegin(ctx) / timeEnd(ctx) -> ms */
static napi_value n_time_begin(napi_env env, napi_callback_info info) {
    napi_value a[1];
    if (!get_args(env, info, 1, a)) return NULL;
    rfx_ctx *c = get_ctx(env, a[0]);
    if (!c) return NULL;
    int rc = rfx_time_begin(c);
    if (rc) return throw_rfx(env, c, "rfx_time_begin", rc);
    return NULL;
}
static napi_value n_time_end(napi_env env, napi_callback_info info) {
    napi_value a[1], out;
    if (!get_args(env, info, 1, a)) return NULL;
    rfx_ctx *c = get_ctx(env, a[0]);
    if (!c) return NULL;
    float ms = 0;
    int rc = rfx_time_end(c, &ms);
    if (rc) return throw_rfx(env, c, "rfx_time_end", rc);
    NAPI_CALL(env, napi_create_double(env, ms, &out));
    return out;
}

/* profile(ctx, enable) / profileRead(ctx) -> { ms: Float64Array(RFX_PROF_COUNT), launches: Int32Array-like array } (rfx_profile, rfx_profile_read) */
static napi_value n_profile(napi_env env, napi_callback_info info) {
    napi_value a[2];
    if (!get_args(env, info, 2, a)) return NULL;
    rfx_ctx *c = get_ctx(env, a[0]);
    if (!c) return NULL;
    int32_t on = 0;
    NAPI_CALL(env, napi_get_value_int32(env, a[1], &on));
    int rc = rfx_profile(c, on);
    if (rc) return throw_rfx(env, c, "rfx_profile", rc);
    return NULL;
}
static napi_value n_profile_read(napi_env env, napi_callback_info info) {
    napi_value a[1], out, ms_arr, n_arr;
    if (!get_args(env, info, 1, a)) return NULL;
    rfx_ctx *c = get_ctx(env, a[0]);
    if (!c) return NULL;
    float ms[RFX_PROF_COUNT];
    int n[RFX_PROF_COUNT];
    int rc = rfx_profile_read(c, ms, n);
    if (rc) return throw_rfx(env, c, "rfx_profile_read", rc);
    NAPI_CALL(env, napi_create_object(env, &out));
    NAPI_CALL(env, napi_create_array_with_length(env, RFX_PROF_COUNT, &ms_arr));
    NAPI_CALL(env, napi_create_array_with_length(env, RFX_PROF_COUNT, &n_arr));
    for (uint32_t i = 0; i < RFX_PROF_COUNT; i++) {
        napi_value v;
        NAPI_CALL(env, napi_create_double(env, ms[i], &v));
        NAPI_CALL(env, napi_set_element(env, ms_arr, i, v));
        NAPI_CALL(env, napi_create_int32(env, n[i], &v));
        NAPI_CALL(env, napi_set_element(env, n_arr, i, v));
    }
    NAPI_CALL(env, napi_set_named_property(env, out, "ms", ms_arr));
    NAPI_CALL(env, napi_set_named_property(env, out, "launches", n_arr));
    return out;
}

static napi_value init(napi_env env, napi_value exports) {
    static const struct { const char *name; napi_callback fn; } fns[] = {
        {"abiVersion", n_abi_version}, {"create", n_create}, {"heldRows", n_held_rows}, {"upload", n_upload}, {"download", n_download},
        {"clear", n_clear}, {"setEnvironment", n_set_environment}, {"setEnvironmentImportance", n_set_environment_importance}, {"packGBuffer", n_pack_gbuffer}, {"packVelocity", n_pack_velocity}, {"ssgiMarch", n_ssgi}, {"ssgiTrace", n_ssgi_trace}, {"ssgiShade", n_ssgi_shade}, {"temporalReproject", n_temporal}, {"copyFramebuffer", n_copy_framebuffer}, {"poissonDenoise", n_denoise}, {"compose", n_compose}, {"finalCompose", n_final},
        {"sync", n_sync}, {"setRowWindow", n_set_row_window}, {"setUvModel", n_set_uv_model}, {"cubeToEquirect", n_cube_to_equirect}, {"haloViolations", n_halo_violations}, {"timeBegin", n_time_begin}, {"timeEnd", n_time_end}, {"profile", n_profile}, {"profileRead", n_profile_read},
        {"stageUpload", n_stage_upload}, {"stageFlip", n_stage_flip}, {"hostAlloc", n_host_alloc},
        {"splitRows", n_split_rows}, {"commUniqueId", n_comm_unique_id}, {"commInit", n_comm_init}, {"haloExchange", n_halo_exchange},
        {"allgatherHistory", n_allgather_history}, {"gatherHistoryRows", n_gather_history_rows}, {"commWait", n_comm_wait}, {"commDestroy", n_comm_destroy},
        {"peerExport", n_peer_export}, {"peerOpen", n_peer_open}, {"peerGatherHistory", n_peer_gather_history}, {"peerClose", n_peer_close}, {"ssgiHitMask", n_ssgi_hit_mask},
    };
    for (size_t i = 0; i < sizeof fns / sizeof fns[0]; i++) {
        napi_value f;
        if (napi_create_function(env, fns[i].name, NAPI_AUTO_LENGTH, fns[i].fn, NULL, &f) != napi_ok) return NULL;
        napi_set_named_property(env, exports, fns[i].name, f);
    }
    return exports;
}

NAPI_MODULE(rfx_napi, init)
