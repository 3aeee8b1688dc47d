#!/usr/bin/env node
"use strict"
// node run_dump.js <dumpdir0> [<dumpdir1> ...] --out <dir> [--steps N --refineSteps N --denoiseIterations N]
// Runs SSGIEffect.update() over a sequence of dumped frames on GPU 0 and writes compose.bin / denoise_b0.bin /
// denoise_b1.bin / temporal0.bin / ssgi.bin / final.bin (the effect's mainImage output) of the LAST frame into --out.
// With --traa '"half"' | '"float"': runs TRAAEffect.update() instead, the dump's direct.bin standing for the composer's
// input buffer (HalfFloatType / FloatType), and writes traa.bin (traa_compose output, RGBA32F) of the last frame.
const fs = require("fs")
const path = require("path")
const rfx = require("./index")

const args = process.argv.slice(2)
const dumps = []
const opt = {}
let out = "."
for (let i = 0; i < args.length; i++) {
	if (args[i] === "--out") out = args[++i]
	else if (args[i].startsWith("--")) opt[args[i].slice(2)] = JSON.parse(args[++i])
	else dumps.push(args[i])
}
if (!dumps.length) {
	console.error("usage: run_dump.js <dumpdir>... --out <dir> [--steps N ...]")
	process.exit(2)
}
const seeds = { ssgi: opt.ssgiSeed === undefined ? 11 : opt.ssgiSeed, denoise: opt.denoiseSeed === undefined ? 22 : opt.denoiseSeed }
delete opt.ssgiSeed
delete opt.denoiseSeed
const first = rfx.readDump(dumps[0])
const scene = { frame: first }
// --env <file.bin> --envWidth W --envHeight H: a raw Float32 RGBA equirect map (row 0 = bottom) as scene.environment (needs --importanceSampling false)
if (opt.env) {
	const b = fs.readFileSync(opt.env)
	scene.environment = { data: new Float32Array(b.buffer, b.byteOffset, b.length / 4), width: opt.envWidth, height: opt.envHeight }
	delete opt.env
	delete opt.envWidth
	delete opt.envHeight
}
const camera = Object.assign({}, first.camera)
const renderer = new rfx.Renderer(first.width, first.height)
if (opt.traa) {
	const half = opt.traa === "half"
	const traa = new rfx.TRAAEffect(scene, camera, new rfx.VelocityDepthNormalPass(scene, camera), { fullAccumulate: true }, true)
	for (const d of dumps) {
		const f = d === dumps[0] ? first : rfx.readDump(d)
		scene.frame = f
		Object.assign(camera, f.camera)
		traa.update(renderer, { texture: { type: half ? rfx.HalfFloatType : rfx.FloatType }, width: f.width, height: f.height, data: f.direct })
	}
	renderer.sync()
	fs.mkdirSync(out, { recursive: true })
	const a = traa.output(renderer)
	fs.writeFileSync(path.join(out, "traa.bin"), Buffer.from(a.buffer, a.byteOffset, a.byteLength))
	console.log(JSON.stringify({ frames: dumps.length, width: first.width, height: first.height, haloViolations: renderer.haloViolations() }))
	process.exit(0)
}
const effect = new rfx.SSGIEffect(null, scene, camera, Object.assign({ width: first.width, height: first.height }, opt), seeds, true)
for (const d of dumps) {
	const f = d === dumps[0] ? first : rfx.readDump(d)
	scene.frame = f
	Object.assign(camera, f.camera)
	effect.update(renderer, null)
}
renderer.sync()
fs.mkdirSync(out, { recursive: true })
const T = rfx.TEX
effect.mainImage(renderer) // the effect's own fragment -> final.bin
for (const [name, tex] of [["final", T.FINAL], ["compose", T.COMPOSE], ["denoise_b0", T.DENOISE_B0], ["denoise_b1", T.DENOISE_B1], ["temporal0", T.TEMPORAL0], ["ssgi", T.SSGI]]) {
	const a = renderer.download(tex)
	fs.writeFileSync(path.join(out, name + ".bin"), Buffer.from(a.buffer, a.byteOffset, a.byteLength))
}
console.log(JSON.stringify({ frames: dumps.length, width: first.width, height: first.height, haloViolations: renderer.haloViolations() }))
