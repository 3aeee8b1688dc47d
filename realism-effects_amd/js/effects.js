"use strict"
// Host side of the hot path in the reference's own language: the reference's class names,
// constructor signatures, option names, defaults and per-frame logic, with every
// `renderer.setRenderTarget(rt); renderer.render(scene, camera)` replaced by one call into
// librfx_hip.so through the N-API addon (Renderer.js).  Node-12 syntax (no ?. ?? class fields).
//
// `scene` / `camera` are plain dumped-state objects (no three.js):
//   camera : { projectionMatrix, projectionMatrixInverse, matrixWorld, matrixWorldInverse (16 numbers, column-major =
//              three's Matrix4.elements), position [3], quaternion [4], near, far, isPerspectiveCamera }
//   scene  : { frame: { depth, gbuffer, velocity, direct } } — the pre-dumped planes (dump.js)
//
// Mirrors: src/ssgi/SSGIEffect.js, src/ssgi/SSGIOptions.js, src/ssgi/pass/SSGIPass.js, src/denoise/Denoiser.js,
// src/denoise/pass/PoissonDenoisePass.js, src/denoise/pass/DenoiserComposePass.js,
// src/temporal-reproject/TemporalReprojectPass.js, src/temporal-reproject/pass/VelocityDepthNormalPass.js,
// src/traa/TRAAEffect.js, src/utils/BlueNoiseUtils.js, src/utils/SceneUtils.js
const { TEX } = require("./Renderer")
const { buildImportance } = require("./envmap")

// src/ssgi/SSGIOptions.js:26-48
const defaultSSGIOptions = {
	mode: "ssgi",
	distance: 10,
	thickness: 10,
	denoiseIterations: 1,
	denoiseKernel: 2,
	denoiseDiffuse: 10,
	denoiseSpecular: 10,
	radius: 3,
	phi: 0.5,
	lumaPhi: 5,
	depthPhi: 2,
	normalPhi: 50,
	roughnessPhi: 50,
	specularPhi: 50,
	envBlur: 0.5,
	importanceSampling: true,
	steps: 20,
	refineSteps: 5,
	resolutionScale: 1,
	missedRays: false,
	outputTexture: null
}

// src/temporal-reproject/TemporalReprojectPass.js:17-32
const defaultTemporalReprojectPassOptions = {
	dilation: false,
	fullAccumulate: false,
	neighborhoodClamp: false,
	neighborhoodClampRadius: 1,
	neighborhoodClampIntensity: 1,
	maxBlend: 1,
	logTransform: false,
	depthDistance: 2,
	worldDistance: 4,
	reprojectSpecular: false,
	renderTarget: null,
	copyTextures: true,
	confidencePower: 0.75,
	inputType: "diffuse"
}

// src/denoise/pass/PoissonDenoisePass.js:16-24
const defaultPoissonBlurOptions = {
	iterations: 1,
	radius: 3,
	phi: 0.5,
	lumaPhi: 5,
	depthPhi: 2,
	normalPhi: 3.25,
	inputType: "diffuseSpecular"
}

// src/denoise/Denoiser.js:6-11
const defaultDenosierOptions = {
	denoiseMode: "full",
	inputType: "diffuseSpecular",
	gBufferPass: null,
	velocityDepthNormalPass: null
}

const INPUT_TYPES = ["diffuseSpecular", "diffuse", "specular"]
const highestSignedInt = 0x7fffffff

// src/utils/BlueNoiseUtils.js:17-33 — the uniform is a getter: every read (one per draw) advances the index
function makeBlueNoiseIndex(startIndex) {
	let blueNoiseIndex = 0
	if (startIndex === undefined || startIndex === null) startIndex = Math.floor(Math.random() * highestSignedInt)
	return {
		startIndex,
		get value() {
			blueNoiseIndex = (startIndex + blueNoiseIndex + 1) % highestSignedInt
			return blueNoiseIndex
		},
		set value(v) {
			blueNoiseIndex = v
		}
	}
}

// src/utils/SceneUtils.js:17-27
function didCameraMove(camera, lastPosition, lastQuaternion) {
	let d2 = 0
	for (let i = 0; i < 3; i++) d2 += (camera.position[i] - lastPosition[i]) * (camera.position[i] - lastPosition[i])
	if (d2 > 0.000001) return true
	const q = camera.quaternion || [0, 0, 0, 1]
	let dot = 0
	for (let i = 0; i < 4; i++) dot += q[i] * lastQuaternion[i]
	const angle = 2 * Math.acos(Math.abs(Math.max(-1, Math.min(1, dot))))
	return angle > 0.001
}

function cloneCamera(camera) {
	return {
		projectionMatrix: Float32Array.from(camera.projectionMatrix),
		projectionMatrixInverse: Float32Array.from(camera.projectionMatrixInverse),
		matrixWorld: Float32Array.from(camera.matrixWorld),
		matrixWorldInverse: Float32Array.from(camera.matrixWorldInverse),
		position: Float32Array.from(camera.position),
		near: camera.near,
		far: camera.far,
		isPerspectiveCamera: camera.isPerspectiveCamera === undefined ? true : camera.isPerspectiveCamera
	}
}

// Stand-in for src/gbuffer/GBufferPass.js: the rasteriser is out of scope; render() hands the pre-dumped
// packed G-buffer + depth planes to the device.
class GBufferPass {
	constructor(scene, camera) {
		this._scene = scene
		this._camera = camera
		this.texture = TEX.GBUFFER
		this.depthTexture = TEX.DEPTH
	}
	setSize(width, height) {
		this.width = width
		this.height = height
	}
	render(renderer) {
		const f = this._scene.frame
		renderer.uploadPlane(TEX.DEPTH, f.depth, f.static)
		if (f.gbuffer) renderer.uploadPlane(TEX.GBUFFER, f.gbuffer, f.static)
		else if (renderer._packedGBuffer !== f.aov) {
			// an engine dump of UNPACKED whole-frame attribute planes: the device packs them (rfx_pack_gbuffer = the pass's fragment epilogue)
			renderer.packGBuffer(Object.assign({ depth: f.depth }, f.aov), 0, renderer.height)
			renderer._packedGBuffer = f.aov
		}
	}
	dispose() {}
}

// src/temporal-reproject/pass/VelocityDepthNormalPass.js:66 — loader shim with the reference's signature
class VelocityDepthNormalPass {
	constructor(scene, camera) {
		this._scene = scene
		this._camera = camera
		this.renderTarget = this
		this.texture = TEX.VELOCITY
		this.depthTexture = TEX.VELOCITY
		this.lastVelocityTexture = null
	}
	setSize(width, height) {
		this.width = width
		this.height = height
	}
	render(renderer) {
		const f = this._scene.frame
		if (f.velocity) renderer.uploadPlane(TEX.VELOCITY, f.velocity, f.static)
		else if (renderer._packedVelocity !== f.aov) {
			renderer.packVelocity({ velocity: f.aov.velocity, normal: f.aov.normal, depth: f.depth }, 0, renderer.height)
			renderer._packedVelocity = f.aov
		}
	}
	dispose() {}
}

// three.js texture `type` constants the path distinguishes (three/src/constants.js)
const UnsignedByteType = 1009
const FloatType = 1015
const HalfFloatType = 1016
function textureType(texture) {
	return texture && texture.type !== undefined && texture.type !== null ? texture.type : FloatType // SSGIPass.js:24 targets are FloatType
}

// value of float16(v) (round to nearest even; overflow -> +-Infinity) as a JS number; Node 12 has no Float16Array
const _f32 = new Float32Array(1)
const _u32 = new Uint32Array(_f32.buffer)
function roundToHalf(v) {
	_f32[0] = v
	const bits = _u32[0]
	const e = (bits >>> 23) & 0xff
	if (e === 0xff || v === 0) return _f32[0]
	const a = Math.abs(_f32[0])
	const q = Math.pow(2, Math.max(e - 127, -14) - 10) // half quantum at this magnitude (subnormals: 2^-24)
	const t = a / q
	const f = Math.floor(t)
	const d = t - f
	let r = (d < 0.5 ? f : d > 0.5 ? f + 1 : f % 2 === 0 ? f : f + 1) * q
	if (r > 65504) r = Infinity
	return bits >>> 31 ? -r : r
}
function toHalfPrecision(data) {
	const out = new Float32Array(data.length)
	for (let i = 0; i < data.length; i++) out[i] = roundToHalf(data[i])
	return out
}

// src/taa/TAAUtils.js:3 + src/temporal-reproject/utils/QuasirandomGenerator.js:12-27
const r2Sequence = (() => {
	const g = 1.32471795724474602596090885447809
	const a1 = 1.0 / g
	const a2 = 1.0 / (g * g)
	const base = 1.1127756842787055
	const points = []
	for (let n = 0; n < 256; n++) points.push([((base + a1 * n) % 1) - 0.5, ((base + a2 * n) % 1) - 0.5])
	return points
})()
// src/taa/TAAUtils.js:5-11 — `setViewOffset` is three's PerspectiveCamera method: a dumped-state camera that has one
// records the offset for the raster side, which renders the next dump with it
function jitter(width, height, camera, frame, jitterScale) {
	if (jitterScale === undefined) jitterScale = 1
	const p = r2Sequence[frame % r2Sequence.length]
	if (camera.setViewOffset) camera.setViewOffset(width, height, p[0] * jitterScale, p[1] * jitterScale, width, height)
	return [p[0] * jitterScale, p[1] * jitterScale]
}

// src/temporal-reproject/TemporalReprojectPass.js:38-225
class TemporalReprojectPass {
	constructor(scene, camera, velocityDepthNormalPass, texture, textureCount, options, halfStoreRTZ) {
		this._scene = scene
		this._camera = camera
		this.textureCount = textureCount
		options = Object.assign({}, defaultTemporalReprojectPassOptions, options || {})
		this.options = options
		this.velocityDepthNormalPass = velocityDepthNormalPass
		this.frame = 0
		this.overrideAccumulatedTextures = []
		this.lastCameraTransform = { position: [0, 0, 0], quaternion: [0, 0, 0, 1] }
		// :63-68 the render target takes the TYPE of the input texture; :137-142 so does the framebuffer copy
		this.targetType = textureType(texture)
		if (this.targetType !== FloatType && this.targetType !== HalfFloatType)
			throw new Error("TemporalReprojectPass: input texture type " + this.targetType + " — only FloatType / HalfFloatType targets are built")
		const it = INPUT_TYPES.indexOf(options.inputType)
		const flags = name => {
			let v = options[name]
			if (!Array.isArray(v)) v = [v, v]
			return [v[0] ? 1 : 0, (v.length > 1 ? v[1] : v[0]) ? 1 : 0]
		}
		this.uniforms = {
			camera: null,
			prevCamera: cloneCamera(camera), // :95-104 the ctor clones the current camera state
			textureCount,
			inputType: it < 0 ? 1 : it,
			reprojectSpecular: flags("reprojectSpecular"),
			neighborhoodClamp: flags("neighborhoodClamp"),
			logTransform: options.logTransform ? 1 : 0,
			fullAccumulate: 0,
			confidencePower: options.confidencePower,
			neighborhoodClampIntensity: options.neighborhoodClampIntensity,
			maxBlend: options.maxBlend,
			keepData: 1,
			historySource: 0,
			inputWidth: 0,
			inputHeight: 0,
			targetHalf: this.targetType === HalfFloatType ? 1 : 0,
			halfStoreRTZ: halfStoreRTZ === undefined || halfStoreRTZ ? 1 : 0
		}
	}
	setSize(width, height) {
		this.width = width
		this.height = height
	}
	get texture() {
		return TEX.TEMPORAL0
	}
	// :137-142 — the slot the pass copies its target into when nothing overrides its history
	get framebufferTexture() {
		return this.targetType === HalfFloatType ? TEX.FBCOPY_F16 : TEX.FBCOPY_F32
	}
	reset() {
		this.uniforms.keepData = 0 // :158-160
	}
	render(renderer) {
		this.frame = (this.frame + 1) % 4096
		const cam = this._camera
		// :168-172,185-187 the pass draws with the UNJITTERED projection (view offset disabled while the uniforms are read)
		this.uniforms.camera = cloneCamera(cam.unjittered || cam)
		const moved = didCameraMove(cam, this.lastCameraTransform.position, this.lastCameraTransform.quaternion)
		this.uniforms.fullAccumulate = this.options.fullAccumulate && !moved ? 1 : 0 // :178-180
		this.lastCameraTransform.position = Array.from(cam.position)
		this.lastCameraTransform.quaternion = Array.from(cam.quaternion || [0, 0, 0, 1])
		const ownHistory = this.overrideAccumulatedTextures.length === 0 // :148-151
		this.uniforms.historySource = !ownHistory ? 0 : this.targetType === HalfFloatType ? 1 : 2
		renderer.temporalReproject(this.uniforms) // :192-193
		this.uniforms.keepData = 1 // :195
		if (ownHistory) {
			renderer.copyFramebuffer(this.framebufferTexture) // :197-201
			if (renderer.afterCopyFramebuffer) renderer.afterCopyFramebuffer(this.framebufferTexture)
		}
		this.uniforms.prevCamera = cloneCamera(cam.unjittered || cam) // :203-213
	}
	jitter(jitterScale) {
		this.unjitter() // :216-220
		return jitter(this.width, this.height, this._camera, this.frame, jitterScale)
	}
	unjitter() {
		if (this._camera.clearViewOffset) this._camera.clearViewOffset() // :222-224
	}
	dispose() {}
}

// src/denoise/pass/PoissonDenoisePass.js:26-152
class PoissonDenoisePass {
	constructor(camera, textures, options, blueNoiseStart, halfStoreRTZ) {
		options = Object.assign({}, defaultPoissonBlurOptions, options || {})
		this.iterations = defaultPoissonBlurOptions.iterations
		this.textures = textures
		let isTextureSpecular = [0, 1]
		if (options.inputType === "diffuse") isTextureSpecular = [0, 0]
		if (options.inputType === "specular") isTextureSpecular = [1, 1]
		this.uniforms = {
			radius: defaultPoissonBlurOptions.radius,
			phi: defaultPoissonBlurOptions.phi,
			lumaPhi: defaultPoissonBlurOptions.lumaPhi,
			depthPhi: options.depthPhi,
			normalPhi: options.normalPhi,
			roughnessPhi: undefined, // :63-64 start undefined until SSGIEffect's setters write them
			specularPhi: undefined,
			textureCount: options.inputType === "diffuseSpecular" ? 2 : 1,
			isTextureSpecular,
			blueNoiseIndex: 0,
			inputIsTemporal: 1,
			writeToB: 0,
			// RGBA16F stores: round-toward-zero like the llvmpipe run of the reference (the parity definition); pass false for
			// round-to-nearest-even like GPU ROPs
			halfStoreRTZ: halfStoreRTZ === undefined || halfStoreRTZ ? 1 : 0
		}
		this.blueNoiseIndex = makeBlueNoiseIndex(blueNoiseStart)
	}
	setSize(width, height) {
		this.width = width
		this.height = height
	}
	get texture() {
		return [TEX.DENOISE_B0, TEX.DENOISE_B1]
	}
	render(renderer) {
		for (let i = 0; i < 2 * this.iterations; i++) {
			const horizontal = i % 2 === 0
			this.uniforms.inputIsTemporal = i === 0 ? 1 : 0
			this.uniforms.writeToB = horizontal ? 0 : 1
			this.uniforms.blueNoiseIndex = this.blueNoiseIndex.value
			renderer.poissonDenoise(this.uniforms)
			if (renderer.afterDenoisePass) renderer.afterDenoisePass(i, this.uniforms)
		}
	}
	dispose() {}
}
PoissonDenoisePass.DefaultOptions = defaultPoissonBlurOptions

// src/denoise/pass/DenoiserComposePass.js:8-136
class DenoiserComposePass {
	constructor(camera, textures, gBufferTexture, depthTexture, options) {
		options = options || {}
		this._camera = camera
		const it = INPUT_TYPES.indexOf(options.inputType)
		// Denoiser.js:53 composerInputTextures: the denoise pass's targets, or (denoiseMode "full_temporal") K2's own
		this.uniforms = { camera: null, inputType: it < 0 ? 0 : it, giSource: textures[0] === TEX.TEMPORAL0 ? 1 : 0 }
	}
	setSize(width, height) {
		this.width = width
		this.height = height
	}
	get texture() {
		return TEX.COMPOSE
	}
	render(renderer) {
		this.uniforms.camera = cloneCamera(this._camera)
		// a row-tiled renderer all-gathers the part of this target K1 reads next frame (.rgb) as 12-byte texels
		this.uniforms.writeHistoryRGB = renderer.gatherHistoryRGB ? 1 : 0
		renderer.compose(this.uniforms)
	}
	dispose() {}
}

// src/denoise/Denoiser.js:16-108
class Denoiser {
	constructor(scene, camera, texture, options, blueNoiseStart, halfStoreRTZ) {
		options = Object.assign({}, defaultDenosierOptions, options || {})
		this.options = options
		this.velocityDepthNormalPass = options.velocityDepthNormalPass || new VelocityDepthNormalPass(scene, camera)
		this.isOwnVelocityDepthNormalPass = !options.velocityDepthNormalPass
		const textureCount = options.inputType === "diffuseSpecular" ? 2 : 1
		const topt = {
			fullAccumulate: true,
			logTransform: true,
			copyTextures: !options.denoise,
			reprojectSpecular: [false, true],
			neighborhoodClamp: [true, true],
			neighborhoodClampRadius: 2,
			neighborhoodClampIntensity: 0.5
		}
		for (const k of Object.keys(defaultTemporalReprojectPassOptions)) if (k in options) topt[k] = options[k]
		this.temporalReprojectPass = new TemporalReprojectPass(scene, camera, this.velocityDepthNormalPass, texture, textureCount, topt)
		this.denoisePass = null
		this.denoiserComposePass = null
		if (options.denoiseMode === "full" || options.denoiseMode === "denoised") {
			const popt = {}
			for (const k of Object.keys(defaultPoissonBlurOptions)) if (k in options) popt[k] = options[k]
			this.denoisePass = new PoissonDenoisePass(camera, [TEX.TEMPORAL0, TEX.TEMPORAL1], popt, blueNoiseStart, halfStoreRTZ)
			this.temporalReprojectPass.overrideAccumulatedTextures = this.denoisePass.texture
		}
		if (["full", "full_temporal", "denoised", "temporal"].indexOf(options.denoiseMode) < 0) throw new Error("denoiseMode " + options.denoiseMode)
		const textures = [TEX.TEMPORAL0, TEX.TEMPORAL1].slice(0, textureCount)
		const composerInputTextures = this.denoisePass ? this.denoisePass.texture : textures // :53
		if (options.denoiseMode.startsWith("full"))
			this.denoiserComposePass = new DenoiserComposePass(camera, composerInputTextures, TEX.GBUFFER, TEX.DEPTH, options)
	}
	get texture() {
		switch (this.options.denoiseMode) {
			case "full":
			case "full_temporal":
				return this.denoiserComposePass.texture
			case "denoised":
				return this.denoisePass.texture
			default:
				return this.temporalReprojectPass.texture
		}
	}
	reset() {
		this.temporalReprojectPass.reset()
	}
	setSize(width, height) {
		for (const p of [this.velocityDepthNormalPass, this.temporalReprojectPass, this.denoisePass, this.denoiserComposePass])
			if (p) p.setSize(width, height)
	}
	dispose() {}
	render(renderer, inputBuffer) {
		if (this.isOwnVelocityDepthNormalPass) this.velocityDepthNormalPass.render(renderer)
		this.temporalReprojectPass.render(renderer)
		if (renderer.afterTemporalPass) renderer.afterTemporalPass()
		if (this.denoisePass) this.denoisePass.render(renderer)
		if (this.denoiserComposePass) {
			this.denoiserComposePass.render(renderer)
			if (renderer.afterComposePass) renderer.afterComposePass()
		}
	}
}

// src/ssgi/pass/SSGIPass.js:7-96
class SSGIPass {
	constructor(ssgiEffect, options, blueNoiseStart) {
		this.ssgiEffect = ssgiEffect
		this._scene = ssgiEffect._scene
		this._camera = ssgiEffect._camera
		this.frame = 21483
		this.uniforms = {
			camera: null,
			steps: 20,
			refineSteps: 5,
			mode: ["ssgi", "ssr"].indexOf(options.mode),
			useDirectLight: 0,
			missedRays: 0,
			importanceSampling: 0,
			useEnvMap: 0,
			rayDistance: 0,
			thickness: 0,
			envBlur: 0,
			blueNoiseIndex: 0
		}
		this.blueNoiseIndex = makeBlueNoiseIndex(blueNoiseStart)
		this.gBufferPass = new GBufferPass(this._scene, this._camera)
	}
	get texture() {
		return TEX.SSGI
	}
	setSize(width, height) {
		// :52-57 the pass's render target (and its `resolution` uniform) is width*resolutionScale x height*resolutionScale
		const s = this.ssgiEffect._options.resolutionScale
		this.renderTargetSize = [width * s, height * s]
		this.uniforms.resolutionScale = s
		this.gBufferPass.setSize(width, height)
	}
	render(renderer) {
		this.frame = (this.frame + 1) % 4096
		this.gBufferPass.render(renderer)
		this.uniforms.camera = cloneCamera(this._camera)
		this.uniforms.blueNoiseIndex = this.blueNoiseIndex.value
		// :89 accumulatedTexture = ssgiEffect.denoiser.texture: K4's target, K2's texture[0] ("temporal"), or — "denoised", where the
		// getter returns the ARRAY of K3's targets — what three binds for a non-texture value: its empty texture (zeros)
		const t = this.ssgiEffect.denoiser.texture
		this.uniforms.historySource = Array.isArray(t) ? 2 : t === TEX.TEMPORAL0 ? 1 : 0
		if (this.uniforms.historySource === 0 && renderer.gatherHistoryRGB) this.uniforms.historySource = 3 // the same values from RFX_TEX_COMPOSE_RGB (tiling.js)
		if (renderer.overlapHistoryGather) {
			// row-tiled run: last frame's composed GI is still being all-gathered; only the shading half of the draw reads it
			renderer.ssgiTrace(this.uniforms)
			renderer.beforeSsgiShade()
			renderer.ssgiShade(this.uniforms)
		} else renderer.ssgiMarch(this.uniforms) // :93-94
	}
	dispose() {}
}

// src/ssgi/SSGIEffect.js:27-439
// three.js texture filter constants (three/src/constants.js)
const NearestFilter = 1003
const LinearFilter = 1006
const LinearMipmapLinearFilter = 1008

// src/ssgi/pass/CubeToEquirectEnvPass.js: scene.environment given as a CubeTexture is rendered into an equirectangular FloatType target (one
// textureCube lookup per texel), read back, and continues as a DataTexture.  The cube as dumped state: { isCubeTexture: true,
// faces: Float32Array(6 * size * size * 4) linear values (+X -X +Y -Y +Z -Z, row j = t as uploaded), size, and the two sampler fields the
// lookup depends on: minFilter (default LinearMipmapLinearFilter, three's Texture default) and generateMipmaps (default true) }.
class CubeToEquirectEnvPass {
	generateEquirectEnvMap(renderer, cubeMap, width = null, height = null, maxWidth = 4096) {
		if (width === null && height === null) {
			// :62-69
			const w = cubeMap.size
			width = 2 ** Math.ceil(Math.log2(2 * w * 3 ** 0.5))
			height = 2 ** Math.ceil(Math.log2(w * 3 ** 0.5))
		}
		if (width > maxWidth) {
			// :71-74
			width = maxWidth
			height = maxWidth / 2
		}
		const minFilter = cubeMap.minFilter === undefined || cubeMap.minFilter === null ? LinearMipmapLinearFilter : cubeMap.minFilter
		let mips
		if (minFilter === LinearMipmapLinearFilter) {
			if (cubeMap.generateMipmaps === false) throw new Error("CubeToEquirectEnvPass: a LinearMipmapLinearFilter cube texture without generated mipmaps is incomplete (samples black)")
			mips = true
		} else if (minFilter === LinearFilter) {
			mips = false
		} else {
			throw new Error("CubeToEquirectEnvPass: cube minFilter " + minFilter + " — LinearFilter and LinearMipmapLinearFilter are built")
		}
		// render + readRenderTargetPixels :76-85, then the DataTexture of :87-97 (FloatType, ClampToEdge, EquirectangularReflectionMapping)
		const data = renderer.cubeToEquirect(cubeMap.faces, cubeMap.size, width, height, mips)
		return { data, width, height, type: FloatType, isCubeTexture: false }
	}
	dispose() {}
}

class SSGIEffect {
	// `seeds` ({ ssgi, denoise } blue-noise start indices) and `halfStoreRTZ` are additions for reproducible offline runs
	constructor(composer, scene, camera, options, seeds, halfStoreRTZ) {
		options = Object.assign({}, defaultSSGIOptions, options || {})
		this._scene = scene
		this._camera = camera
		this.composer = composer
		this.isUsingRenderPass = true
		if (options.mode === "ssr") {
			options.reprojectSpecular = true // :70-73
			options.neighborhoodClamp = true
			options.inputType = "specular"
		} else if (options.mode === "ssgi") {
			options.reprojectSpecular = [false, true] // :74-77
			options.neighborhoodClamp = [false, true]
		}
		if (typeof options.preset === "string") {
			// :79-99 (the second `case "medium"` in the reference is unreachable)
			switch (options.preset) {
				case "low":
					options.steps = 10
					options.refineSteps = 2
					options.denoiseMode = "full_temporal"
					break
				case "medium":
					options.steps = 20
					options.refineSteps = 4
					options.denoiseMode = "full"
					break
			}
		}
		seeds = seeds || {}
		this._options = options
		this._halfStoreRTZ = halfStoreRTZ
		this.ssgiPass = new SSGIPass(this, options, seeds.ssgi)
		this.denoiser = new Denoiser(
			scene,
			camera,
			this.ssgiPass.texture,
			Object.assign({ gBufferPass: this.ssgiPass.gBufferPass, velocityDepthNormalPass: options.velocityDepthNormalPass }, options),
			seeds.denoise,
			halfStoreRTZ
		)
		this.lastSize = { width: options.width, height: options.height, resolutionScale: options.resolutionScale }
		// FinalSSGIMaterial uniforms (:47-66)
		this.uniforms = { camera: cloneCamera(camera), isDebug: 0, inputSource: 0, fogMode: 0, fogColor: [0, 0, 0], fogNear: 0, fogFar: 0, fogDensity: 0 }
		this.setSize(options.width, options.height)
		this.makeOptionsReactive(options)
		this.outputTexture = this.denoiser.texture
		this.updateUsingRenderPass() // the composer's RenderPass ran before update(): direct light is available
	}

	updateUsingRenderPass() {
		this.ssgiPass.uniforms.useDirectLight = this.isUsingRenderPass ? 1 : 0 // :143-151
	}

	reset() {
		this.denoiser.reset()
	}

	// :157-268
	makeOptionsReactive(options) {
		let needsUpdate = false
		const ssgiUniforms = this.ssgiPass.uniforms
		for (const key of Object.keys(options)) {
			if (key === "outputTexture") continue
			Object.defineProperty(this, key, {
				configurable: true,
				get() {
					return options[key]
				},
				set(value) {
					if (options[key] === value && needsUpdate) return
					options[key] = value
					const dp = this.denoiser.denoisePass
					switch (key) {
						case "denoiseIterations":
							if (dp) dp.iterations = value
							break
						case "radius":
						case "phi":
						case "lumaPhi":
						case "depthPhi":
						case "normalPhi":
						case "roughnessPhi":
						case "specularPhi":
							if (dp) {
								dp.uniforms[key] = value
								this.reset()
							}
							break
						case "resolutionScale":
							this.setSize(this.lastSize.width, this.lastSize.height)
							this.reset()
							break
						case "steps":
						case "refineSteps":
							ssgiUniforms[key] = parseInt(value)
							this.reset()
							break
						case "importanceSampling":
							// only effective with an env map (SSGIEffect.js:344-354): keepEnvMapUpdated re-reads the option when the environment is (re)set
							this._envUuid = null
							this.reset()
							break
						case "missedRays":
							ssgiUniforms.missedRays = value ? 1 : 0
							this.reset()
							break
						case "distance":
							ssgiUniforms.rayDistance = value
							this.reset()
							break
						default:
							// must be a uniform (:254-258); denoiseKernel/denoiseDiffuse/denoiseSpecular have no consumer
							if (key === "thickness" || key === "envBlur") {
								ssgiUniforms[key] = value
								this.reset()
							}
					}
				}
			})
			this[key] = options[key]
		}
		needsUpdate = true
	}

	setSize(width, height, force) {
		if (width === undefined && height === undefined) return
		this.ssgiPass.setSize(width, height)
		this.denoiser.setSize(width, height)
		// K2 samples the pass's (possibly smaller) texture NEAREST at full-resolution vUv: the device needs its size
		const tu = this.denoiser.temporalReprojectPass.uniforms
		const scaled = this._options.resolutionScale !== 1
		tu.inputWidth = scaled ? Math.trunc(this.ssgiPass.renderTargetSize[0]) : 0
		tu.inputHeight = scaled ? Math.trunc(this.ssgiPass.renderTargetSize[1]) : 0
		this.lastSize = { width, height, resolutionScale: this._options.resolutionScale }
	}

	get depthTexture() {
		return this.ssgiPass.gBufferPass.depthTexture
	}

	initialize() {}

	dispose() {
		this.ssgiPass.dispose()
		this.denoiser.dispose()
	}

	// :372-436.  inputBuffer: the composer's input buffer (direct lighting) as a Float32Array RGBA plane, or null to
	// take scene.frame.direct.
	// :309-362.  scene.environment: null/undefined, or an equirectangular HDR map as dumped state — { data: Float32Array(H*W*4, row 0 =
	// bottom), width, height, type: HalfFloatType (RGBELoader's, the default) | FloatType }.  The effect turns its mipmaps on (:323-328):
	// here the device builds the chain (rfx_set_environment).
	keepEnvMapUpdated(renderer) {
		const env = this._scene.environment
		const u = this.ssgiPass.uniforms
		if (env) {
			if (this._envUuid !== env) {
				let map = env
				if (map.isCubeTexture) {
					// :316-321 convert it to an equirectangular texture so the pass can sample it and use MIS
					if (!this.cubeToEquirectEnvPass) this.cubeToEquirectEnvPass = new CubeToEquirectEnvPass()
					map = this.cubeToEquirectEnvPass.generateEquirectEnvMap(renderer, map)
				}
				const half = map.type === undefined || map.type === null || map.type === HalfFloatType
				renderer.setEnvironment(map.data, map.width, map.height, half, this._halfStoreRTZ === undefined || this._halfStoreRTZ)
				u.importanceSampling = 0
				if (this._options.importanceSampling) {
					// :348-351 EquirectHdrInfoUniform.updateFrom, then the define.  The worker sees the half-float texels (fromHalfFloat); `data` is
					// in GL row order (row 0 = bottom): with texture.flipY the reference's array is the other way up and the worker "un-flips" it
					let texels = half ? toHalfPrecision(map.data) : map.data
					if (map.flipY) {
						const r = new Float32Array(texels.length)
						const rowLen = map.width * 4
						for (let y = 0; y < map.height; y++) r.set(texels.subarray(y * rowLen, (y + 1) * rowLen), (map.height - 1 - y) * rowLen)
						texels = r
					}
					const imp = buildImportance(texels, map.width, map.height, !!map.flipY)
					renderer.setEnvironmentImportance(imp.marginalWeights, imp.conditionalWeights, imp.totalSumValue)
					u.importanceSampling = 1
				}
				this._envUuid = env
				u.useEnvMap = 1 // defines.USE_ENVMAP :344
				this.reset() // :356
			}
		} else if (u.useEnvMap) {
			u.useEnvMap = 0 // :361-366
			u.importanceSampling = 0
			renderer.setEnvironment(null)
			this._envUuid = null
		}
	}

	update(renderer, inputBuffer) {
		this.keepEnvMapUpdated(renderer)
		const direct = inputBuffer || this._scene.frame.direct
		renderer.uploadPlane(TEX.DIRECT_LIGHT, direct, this._scene.frame && this._scene.frame.static)
		this.ssgiPass.render(renderer)
		this.denoiser.render(renderer, inputBuffer)
		// :400-417 the effect's own uniforms: inputTexture = the denoiser's texture, sceneTexture = the input buffer, fog from the scene
		const fog = this._scene.fog
		const u = this.uniforms
		let out = this.denoiser.texture // :139,402 inputTexture = outputTexture[0] ?? outputTexture
		if (Array.isArray(out)) out = out[0]
		u.inputSource = out === TEX.COMPOSE ? 0 : out === TEX.TEMPORAL0 ? 1 : 2
		u.fogMode = !fog ? 0 : fog.isFogExp2 ? 2 : 1
		if (fog) {
			u.fogColor = Array.from(fog.color)
			u.fogNear = fog.near || 0
			u.fogFar = fog.far || 0
			u.fogDensity = fog.density || 0
			u.camera = cloneCamera(this._camera)
		}
	}
	// The effect's own fragment (src/ssgi/shader/ssgi_compose.frag:20-45), which postprocessing's EffectPass runs after update():
	// scene colour on background texels, composed GI (+ fog) elsewhere, alpha 1 -> TEX.FINAL
	mainImage(renderer) {
		renderer.finalCompose(this.uniforms)
		return TEX.FINAL
	}
}
SSGIEffect.DefaultOptions = defaultSSGIOptions

// src/ssgi/SSREffect.js:3-9
class SSREffect extends SSGIEffect {
	constructor(composer, scene, camera, options, seeds, halfStoreRTZ) {
		options = Object.assign({}, options || {})
		options.mode = "ssr"
		super(composer, scene, camera, options, seeds, halfStoreRTZ)
	}
}

// src/traa/TRAAEffect.js:10-78 — option surface + K2 parameter mapping (camera jitter needs the rasteriser)
class TRAAEffect {
	// `halfStoreRTZ` is an addition for parity runs against the llvmpipe oracle (RGBA16F stores truncate there)
	constructor(scene, camera, velocityDepthNormalPass, options, halfStoreRTZ) {
		this._scene = scene
		this._camera = camera
		this.velocityDepthNormalPass = velocityDepthNormalPass
		options = Object.assign({}, options || defaultTemporalReprojectPassOptions, {
			maxBlend: 0.9,
			neighborhoodClamp: true,
			neighborhoodClampIntensity: 1,
			neighborhoodClampRadius: 1,
			logTransform: true,
			confidencePower: 4
		})
		this.options = Object.assign({}, defaultTemporalReprojectPassOptions, options)
		this.temporalReprojectPass = null
		this._halfStoreRTZ = halfStoreRTZ
		this.uniforms = { accumulatedTexture: null }
		this.unjitteredProjectionMatrix = null
	}
	setSize(width, height) {
		if (this.temporalReprojectPass) this.temporalReprojectPass.setSize(width, height)
	}
	reset() {
		this.temporalReprojectPass.reset()
	}
	temporalParams(texture) {
		if (!this.temporalReprojectPass)
			this.temporalReprojectPass = new TemporalReprojectPass(this._scene, this._camera, this.velocityDepthNormalPass, texture, 1, this.options, this._halfStoreRTZ)
		return this.temporalReprojectPass.uniforms
	}
	// `inputBuffer`: the composer buffer as dumped state — { texture: { type }, width, height, data: Float32Array(H*W*4) }
	update(renderer, inputBuffer) {
		if (!this.temporalReprojectPass) {
			this.temporalParams(inputBuffer.texture) // :53-66
			this.temporalReprojectPass.setSize(inputBuffer.width, inputBuffer.height)
			this.uniforms.accumulatedTexture = this.temporalReprojectPass.texture
		}
		// the raster shims stand where the composer's earlier passes ran
		this.velocityDepthNormalPass.render(renderer)
		let data = inputBuffer.data
		if (this.temporalReprojectPass.targetType === HalfFloatType) {
			// a HalfFloatType buffer holds half-precision texels: state that on the way in (exact for a real dump of one)
			if (this._halfSrc !== data) {
				this._halfSrc = data
				this._halfData = toHalfPrecision(data)
			}
			data = this._halfData
		}
		renderer.uploadPlane(TEX.SSGI, new Uint32Array(data.buffer, data.byteOffset, data.length), false) // K2's `inputTexture` (:118)
		this.temporalReprojectPass.unjitter() // :68-73
		this.unjitteredProjectionMatrix = Array.from(this._camera.projectionMatrix)
		this.temporalReprojectPass.jitter()
		this.temporalReprojectPass.render(renderer) // :75
	}
	// traa_compose.frag:3-7 — outputColor = vec4(accumulatedTexel.rgb, 1.)
	output(renderer, row0, rows) {
		const t = renderer.download(this.uniforms.accumulatedTexture, row0, rows)
		for (let i = 3; i < t.length; i += 4) t[i] = 1
		return t
	}
	dispose() {
		if (this.temporalReprojectPass) this.temporalReprojectPass.dispose()
	}
}
TRAAEffect.DefaultOptions = defaultTemporalReprojectPassOptions

module.exports = {
	SSGIEffect,
	CubeToEquirectEnvPass,
	NearestFilter,
	LinearFilter,
	LinearMipmapLinearFilter,
	SSREffect,
	TRAAEffect,
	VelocityDepthNormalPass,
	TemporalReprojectPass,
	PoissonDenoisePass,
	DenoiserComposePass,
	Denoiser,
	SSGIPass,
	GBufferPass,
	defaultSSGIOptions,
	defaultTemporalReprojectPassOptions,
	defaultPoissonBlurOptions,
	FloatType,
	HalfFloatType,
	UnsignedByteType,
	roundToHalf,
	r2Sequence,
	jitter,
	makeBlueNoiseIndex,
	didCameraMove
}
