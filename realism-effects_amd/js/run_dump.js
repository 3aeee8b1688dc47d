#!/usr/bin/env node
"use strict"
// node run_dump.js <dumpdir0> [<dumpdir1> ...] --out <dir> [--steps N --refineSteps N --denoiseIterations N]
// Runs SSGIEffect.update() over a sequence of dumped frames on GPU 0 and writes compose.bin / denoise_b0.bin /
// denoise_b1.bin / temporal0.bin / ssgi.bin / final.bin (the effect's mainImage output) of the LAST frame into --out.
// With --traa '"half"' | '"float"': runs TRAAEffect.update() instead, the dump's direct.bin standing for the composer's
// input buffer (HalfFloatType / FloatType), and writes traa.bin (traa_compose output, RGBA32F) of the last frame.
// --png <file> [--tonemap '"aces"'|'"linear"' --exposure X] / --exr <file> / --pfm <file>: also write final.bin as an image (tone-mapped
// 8-bit sRGB PNG; scene-linear float OpenEXR / PFM) — js/imageio.js.
// --uvModel '"reference_gl"': the reference GL's own vUv instead of (i + 0.5) / n (parity runs against the reference on llvmpipe).
// --stream true: the dumps cross PCIe on the context's upload stream from pinned planes, frame n+1 while frame n is drawn
// (rfx_stage_upload / rfx_stage_flip); same outputs.
// With --ranks N (N > 1): the frame is cut into N row tiles, ONE NODE PROCESS PER GPU (this process spawns them: rank r drives
// device r), which exchange halo rows and the composed GI over RCCL through the C ABI (js/tiling.js); rank 0 creates the
// ncclUniqueId and hands it over through a file.  The parent stitches the tiles: the outputs are bit-identical to a --ranks 1 run.
// --historyGather '"all"' (default) | '"bounded"' | '"peer"': how next frame's K1 gets the composed GI of the other tiles (js/tiling.js); "peer"
// moves no collective at all — each rank's kernel loads what its rays read through HIP IPC mappings (the blobs travel through files too).
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
// ---- row-tiled run: parent process
if (opt.ranks > 1 && opt.rank === undefined) {
	const cp = require("child_process")
	const os = require("os")
	fs.mkdirSync(out, { recursive: true })
	const idDir = fs.mkdtempSync(path.join(os.tmpdir(), "rfx-"))
	const idFile = path.join(idDir, "nccl_id")
	const cleanup = () => {
		try { fs.unlinkSync(idFile) } catch (e) { /* never written */ }
		for (let r = 0; r < opt.ranks; r++) try { fs.unlinkSync(idFile + ".peer" + r) } catch (e) { /* not that mode */ }
		try { fs.rmdirSync(idDir) } catch (e) { /* not empty: leave it */ }
	}
	const kids = []
	for (let r = 0; r < opt.ranks; r++)
		kids.push(cp.spawn(process.execPath, [__filename].concat(args, ["--rank", String(r), "--idFile", JSON.stringify(idFile)]), { stdio: ["ignore", "pipe", "inherit"] }))
	let left = kids.length, failed = false
	const lines = new Array(kids.length).fill("")
	kids.forEach((k, r) => {
		k.stdout.on("data", d => (lines[r] += d))
		k.on("exit", code => {
			if (code !== 0 && !failed) {
				// a rank that dies before the communicator exists leaves the others waiting for it (ncclCommInitRank, the id file): stop them now
				failed = true
				console.error("rank " + r + " exited with code " + code + ": stopping the other ranks")
				kids.forEach((q, i) => { if (i !== r && q.exitCode === null) q.kill() })
			}
			if (--left) return
			cleanup()
			if (failed) process.exit(1)
			// stitch the row tiles (rank order = ascending rows)
			for (const name of ["final", "compose", "denoise_b0", "denoise_b1", "temporal0", "ssgi"]) {
				const parts = kids.map((_, q) => fs.readFileSync(path.join(out, name + ".rank" + q + ".bin")))
				fs.writeFileSync(path.join(out, name + ".bin"), Buffer.concat(parts))
				kids.forEach((_, q) => fs.unlinkSync(path.join(out, name + ".rank" + q + ".bin")))
			}
			const info = lines.map(l => JSON.parse(l.trim().split("\n").pop()))
			console.log(JSON.stringify({ frames: dumps.length, width: info[0].width, height: info[0].height, ranks: opt.ranks, haloRows: info[0].haloRows,
				haloViolations: info.reduce((a, b) => a + b.haloViolations, 0), exchanges: info[0].exchanges }))
		})
	})
	return
}
const tiled = opt.ranks > 1 ? { rank: opt.rank, ranks: opt.ranks, idFile: opt.idFile, historyGather: opt.historyGather || "all" } : null
delete opt.historyGather
delete opt.ranks
delete opt.rank
delete opt.idFile
const stream = !!opt.stream
delete opt.stream
const images = { png: opt.png, exr: opt.exr, pfm: opt.pfm, tonemap: opt.tonemap, exposure: opt.exposure }
for (const k of Object.keys(images)) delete opt[k]
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
// --envCube <file.bin> --envCubeSize S [--envCubeMipmaps false]: scene.environment as a CubeTexture — six S x S Float32 RGBA faces (+X -X +Y -Y
// +Z -Z, row j = t as uploaded), converted once through CubeToEquirectEnvPass (three's default sampler state unless --envCubeMipmaps false:
// LinearFilter, no chain)
if (opt.envCube) {
	const b = fs.readFileSync(opt.envCube)
	scene.environment = { isCubeTexture: true, faces: new Float32Array(b.buffer, b.byteOffset, b.length / 4), size: opt.envCubeSize }
	if (opt.envCubeMipmaps === false) Object.assign(scene.environment, { minFilter: rfx.LinearFilter, generateMipmaps: false })
	delete opt.envCube
	delete opt.envCubeSize
	delete opt.envCubeMipmaps
}
const camera = Object.assign({}, first.camera)
let renderer
if (tiled) {
	// halo: the K3 tap footprint and the largest vertical motion of the sequence (rfx_amd/tiling.py required_halo)
	let vmax = 0
	for (const d of dumps) {
		const f = d === dumps[0] ? first : rfx.readDump(d)
		if (!f.velocity) throw new Error("--ranks needs packed velocity.bin dumps")
		const v = new Float32Array(f.velocity.buffer, f.velocity.byteOffset, f.velocity.length)
		for (let i = 1; i < v.length; i += 4) if (Math.abs(v[i]) > vmax) vmax = Math.abs(v[i])
	}
	const halo = rfx.requiredHalo(opt.radius === undefined ? 3 : opt.radius, vmax, first.height, first.width)
	const waitFor = (file, what) => {
		const t0 = Date.now()
		while (!fs.existsSync(file)) {
			if (Date.now() - t0 > 120000) throw new Error("rank " + tiled.rank + ": no " + what)
			Atomics.wait(new Int32Array(new SharedArrayBuffer(4)), 0, 0, 20)
		}
		return fs.readFileSync(file)
	}
	let id
	if (tiled.rank === 0) {
		id = rfx.commUniqueId()
		fs.writeFileSync(tiled.idFile + ".tmp", id)
		fs.renameSync(tiled.idFile + ".tmp", tiled.idFile)
	} else {
		id = waitFor(tiled.idFile, "ncclUniqueId from rank 0")
	}
	renderer = new rfx.TiledRenderer(first.width, first.height, tiled.rank, tiled.ranks, halo, id,
		{ device: process.env.RFX_ONE_GPU === "1" ? 0 : tiled.rank, historyGather: tiled.historyGather === "peer" ? "all" : tiled.historyGather })
	// --historyGather '"peer"': the ranks' export blobs (192 plain bytes each) travel once, through files next to the id file
	if (tiled.historyGather === "peer")
		renderer.usePeerHistory(blob => {
			fs.writeFileSync(tiled.idFile + ".peer" + tiled.rank + ".tmp", blob)
			fs.renameSync(tiled.idFile + ".peer" + tiled.rank + ".tmp", tiled.idFile + ".peer" + tiled.rank)
			const all = []
			for (let r = 0; r < tiled.ranks; r++) all.push(waitFor(tiled.idFile + ".peer" + r, "peer blob of rank " + r))
			return all
		})
} else renderer = new rfx.Renderer(first.width, first.height)
// --uvModel '"reference_gl"': every fragment sees the vUv the reference GL's rasteriser interpolates (rfx_set_uv_model) instead of (i + 0.5) / n
if (opt.uvModel) (renderer.inner || renderer).setUvModel(opt.uvModel)
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
if (stream && !tiled) {
	if (!first.gbuffer) throw new Error("--stream needs packed gbuffer.bin / velocity.bin dumps")
	const n = first.width * first.height
	const sets = [0, 1].map(() => ({ depth: rfx.Renderer.hostAlloc(Float32Array, n), gbuffer: rfx.Renderer.hostAlloc(Uint32Array, 4 * n),
		velocity: rfx.Renderer.hostAlloc(Uint32Array, 4 * n), direct: rfx.Renderer.hostAlloc(Float32Array, 4 * n) }))
	const load = (d, set) => { // disk -> pinned planes (a reader thread's job in a long run)
		const f = d === dumps[0] ? first : rfx.readDump(d)
		for (const k of ["depth", "gbuffer", "velocity", "direct"]) set[k].set(f[k])
		return Object.assign({}, f, set, { static: "resident" })
	}
	let cur = load(dumps[0], sets[0])
	renderer.stageFrame(cur)
	renderer.stageFlip()
	for (let i = 0; i < dumps.length; i++) {
		const next = i + 1 < dumps.length ? load(dumps[i + 1], sets[(i + 1) & 1]) : null
		if (next) renderer.stageFrame(next) // frame i+1 starts crossing PCIe ...
		scene.frame = cur
		Object.assign(camera, cur.camera)
		effect.update(renderer, null) // ... while frame i is drawn
		renderer.stageFlip()
		cur = next
	}
} else
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
	// a tile writes its own rows; the parent stitches them
	const a = tiled ? renderer.download(tex, renderer.tileY0, renderer.tileRows) : renderer.download(tex)
	fs.writeFileSync(path.join(out, name + (tiled ? ".rank" + tiled.rank : "") + ".bin"), Buffer.from(a.buffer, a.byteOffset, a.byteLength))
}
if (!tiled && (images.png || images.exr || images.pfm)) {
	const fin = renderer.download(T.FINAL)
	if (images.exr) rfx.writeEXR(images.exr, fin, first.width, first.height)
	if (images.pfm) rfx.writePFM(images.pfm, fin, first.width, first.height)
	if (images.png) rfx.writePNG(images.png, rfx.tonemap(fin, first.width, first.height, images.tonemap, images.exposure), first.width, first.height, 3)
}
console.log(JSON.stringify({ frames: dumps.length, width: first.width, height: first.height, haloViolations: renderer.haloViolations(),
	haloRows: tiled ? renderer.haloRows : 0, exchanges: tiled ? renderer.exchangeCount : 0 }))
