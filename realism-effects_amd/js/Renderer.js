"use strict"
// Renderer — the device.  Stands where three's WebGLRenderer stands in the reference's
// `pass.render(renderer)` calls: one Renderer = one rfx context (librfx_hip.so) on one GPU, or on one
// row tile of the frame (tileY0 / tileRows / haloRows).  Thin wrapper over the N-API addon
// (../napi/rfx_napi.node); there is NO software fallback: loading fails if the addon or the HIP
// library is missing, and creation fails without an MI355X.
const fs = require("fs")
const path = require("path")
const addon = require("../napi/rfx_napi.node")

// texture slots — include/rfx.h `rfx_tex`
const TEX = {
	DEPTH: 0,
	GBUFFER: 1,
	VELOCITY: 2,
	DIRECT_LIGHT: 3,
	BLUE_NOISE: 4,
	SSGI: 5,
	TEMPORAL0: 6,
	TEMPORAL1: 7,
	DENOISE_A0: 8,
	DENOISE_A1: 9,
	DENOISE_B0: 10,
	DENOISE_B1: 11,
	COMPOSE: 12,
	FBCOPY_F16: 13,
	FBCOPY_F32: 14,
	FINAL: 15,
	COMPOSE_RGB: 16
}
// [TypedArray constructor, elements per texel]
const FORMAT = {
	0: [Float32Array, 1],
	1: [Uint32Array, 4],
	2: [Uint32Array, 4],
	3: [Float32Array, 4],
	4: [Uint8Array, 4],
	5: [Uint32Array, 4],
	6: [Float32Array, 4],
	7: [Float32Array, 4],
	8: [Uint16Array, 4],
	9: [Uint16Array, 4],
	10: [Uint16Array, 4],
	11: [Uint16Array, 4],
	12: [Float32Array, 4],
	13: [Uint16Array, 4],
	14: [Float32Array, 4],
	15: [Float32Array, 4],
	16: [Float32Array, 3]
}

// 128x128 RGBA8 blue-noise table: decoded once from the reference's PNG asset, already flipY'd
// (tools/make_blue_noise_table.py; src/utils/BlueNoiseUtils.js:9-15)
function loadBlueNoiseTable() {
	const buf = fs.readFileSync(path.join(__dirname, "..", "data", "blue_noise_128_rgba8.bin"))
	if (buf.length !== 128 * 128 * 4) throw new Error("blue noise table: unexpected size")
	return new Uint8Array(buf.buffer, buf.byteOffset, buf.length)
}

class Renderer {
	constructor(width, height, options) {
		options = options || {}
		this.width = width
		this.height = height
		this.tileY0 = options.tileY0 || 0
		this.tileRows = options.tileRows === undefined ? height - this.tileY0 : options.tileRows
		this.haloRows = options.haloRows || 0
		this._h = addon.create(options.device || 0, width, height, this.tileY0, this.tileRows, this.haloRows)
		this._resident = {}
		this.upload(TEX.BLUE_NOISE, loadBlueNoiseTable(), 0, 128)
	}

	heldRows(tex) {
		return addon.heldRows(this._h, tex)
	}

	// rows [row0, row0+rows) in FRAME rows; `array` holds exactly those rows
	upload(tex, array, row0, rows) {
		const held = this.heldRows(tex)
		if (row0 === undefined) row0 = held[0]
		if (rows === undefined) rows = held[1]
		addon.upload(this._h, tex, array, row0, rows)
	}

	// a dumped FULL-FRAME plane: the slot takes the band it holds.  Every call uploads (the reference re-renders its raster passes every
	// frame) unless the dump declares itself unchanged (frame.static): then an already resident plane object is not re-sent — opt-in,
	// because a buffer refilled in place is the same object with new texels
	uploadPlane(tex, plane, isStatic) {
		if (isStatic === "resident" || (isStatic && this._resident[tex] === plane)) return
		const held = this.heldRows(tex)
		const per = FORMAT[tex][1] * this.width
		const band = plane.length === held[1] * per ? plane : plane.subarray(held[0] * per, (held[0] + held[1]) * per)
		this.upload(tex, band, held[0], held[1])
		this._resident[tex] = plane
	}

	// streaming dumps (rfx.h): stage the NEXT frame's planes asynchronously (pinned memory from Renderer.hostAlloc), then stageFlip()
	// after the current frame's draws; a frame streamed this way carries `static: "resident"` so the loader shims do not upload it again
	static hostAlloc(Ctor, length) {
		return new Ctor(addon.hostAlloc(length * Ctor.BYTES_PER_ELEMENT))
	}
	stageFrame(frame) {
		for (const pt of [[TEX.DEPTH, frame.depth], [TEX.GBUFFER, frame.gbuffer], [TEX.VELOCITY, frame.velocity], [TEX.DIRECT_LIGHT, frame.direct]]) {
			const held = this.heldRows(pt[0])
			const per = FORMAT[pt[0]][1] * this.width
			const plane = pt[1]
			const band = plane.length === held[1] * per ? plane : plane.subarray(held[0] * per, (held[0] + held[1]) * per)
			addon.stageUpload(this._h, pt[0], band, held[0], held[1])
		}
		this._staged = frame // keep the planes alive while they are in flight
	}
	stageFlip() {
		addon.stageFlip(this._h)
	}

	download(tex, row0, rows) {
		const held = this.heldRows(tex)
		if (row0 === undefined) row0 = held[0]
		if (rows === undefined) rows = held[1]
		const f = FORMAT[tex]
		const out = new f[0](rows * (tex === TEX.BLUE_NOISE ? 128 : this.width) * f[1])
		addon.download(this._h, tex, out, row0, rows)
		return out
	}

	clear(tex) {
		addon.clear(this._h, tex)
	}

	// importer: unpacked attribute planes (Float32Arrays over rows [row0, row0+rows)) -> the packed render targets (rfx_pack_gbuffer / _velocity)
	packGBuffer(aov, row0, rows) {
		const held = this.heldRows(TEX.GBUFFER)
		addon.packGBuffer(this._h, aov, row0 === undefined ? held[0] : row0, rows === undefined ? held[1] : rows)
	}
	packVelocity(aov, row0, rows) {
		const held = this.heldRows(TEX.VELOCITY)
		addon.packVelocity(this._h, aov, row0 === undefined ? held[0] : row0, rows === undefined ? held[1] : rows)
	}

	// scene.environment: Float32Array(H*W*4) equirect map (row 0 = bottom) or null — rfx_set_environment
	setEnvironment(data, width, height, halfFloatType, halfStoreRTZ) {
		addon.setEnvironment(this._h, data || null, width || 0, height || 0, halfFloatType ? 1 : 0, halfStoreRTZ ? 1 : 0)
	}

	// EquirectHdrInfoUniform's tables for importanceSampling (js/envmap.js buildImportance) — rfx_set_environment_importance
	setEnvironmentImportance(marginalWeights, conditionalWeights, totalSumValue) {
		const whole = Math.trunc(totalSumValue) // ~~totalSumValue (EquirectHdrInfoUniform.js:391-394)
		addon.setEnvironmentImportance(this._h, marginalWeights, conditionalWeights, whole, totalSumValue - whole)
	}

	// the four draws + the framebuffer copy (include/rfx.h)
	ssgiMarch(uniforms) {
		addon.ssgiMarch(this._h, uniforms)
	}
	// restrict the rows the following draws produce to [y0, y1); no arguments resets (rfx_set_row_window)
	setRowWindow(y0, y1) {
		addon.setRowWindow(this._h, y0 | 0, y1 | 0)
	}
	// CubeToEquirectEnvPass's draw + read-back (rfx_cube_to_equirect): faces = Float32Array(6 * size * size * 4), +X -X +Y -Y +Z -Z, row j = t
	// as uploaded; returns Float32Array(width * height * 4), row 0 = bottom
	cubeToEquirect(faces, size, width, height, generateMipmaps) {
		const out = new Float32Array(width * height * 4)
		addon.cubeToEquirect(this._h, faces, size | 0, generateMipmaps ? 1 : 0, out, width | 0, height | 0)
		return out
	}
	// which vUv the following draws' fragments see (rfx_set_uv_model): "ideal" = (i + 0.5) / n, "reference_gl" = what the reference GL's
	// rasteriser interpolates for three's full-screen triangle, bit for bit
	setUvModel(model) {
		const m = { ideal: 0, reference_gl: 1 }[model]
		if (m === undefined) throw new RangeError("setUvModel: \"ideal\" or \"reference_gl\"")
		addon.setUvModel(this._h, m)
	}
	// the same draw in two launches (rfx_ssgi_trace / rfx_ssgi_shade): only the second reads last frame's composed GI
	ssgiTrace(uniforms) {
		addon.ssgiTrace(this._h, uniforms)
	}
	ssgiShade(uniforms) {
		addon.ssgiShade(this._h, uniforms)
	}
	temporalReproject(uniforms) {
		addon.temporalReproject(this._h, uniforms)
	}
	// renderer.copyFramebufferToTexture of TemporalReprojectPass.js:198-201
	copyFramebuffer(dstTex) {
		addon.copyFramebuffer(this._h, dstTex)
	}
	poissonDenoise(uniforms) {
		addon.poissonDenoise(this._h, uniforms)
	}
	compose(uniforms) {
		addon.compose(this._h, uniforms)
	}

	// SSGIEffect's own fragment (ssgi_compose.frag mainImage)
	finalCompose(uniforms) {
		addon.finalCompose(this._h, uniforms)
	}

	sync() {
		addon.sync(this._h)
	}
	haloViolations() {
		return addon.haloViolations(this._h)
	}
	// per-draw device timing inside a frame loop (rfx_profile / rfx_profile_read): { ms: [...], launches: [...] } indexed by RFX_PROF_* of include/rfx.h
	profile(enable) {
		addon.profile(this._h, enable ? 1 : 0)
	}
	profileRead() {
		return addon.profileRead(this._h)
	}
	timeBegin() {
		addon.timeBegin(this._h)
	}
	timeEnd() {
		return addon.timeEnd(this._h)
	}
	dispose() {
		this._h = null // the context is destroyed by the handle's finalizer
	}
}

module.exports = { Renderer, TEX, FORMAT, loadBlueNoiseTable, abiVersion: addon.abiVersion }
