"use strict"
// Dump directory reader: frame.json + raw little-endian planes (row 0 = bottom).
//   frame.json : { width, height, camera: {...16-number matrices..., position, quaternion, near, far}, prevCamera: {...} }
//   depth.bin    Float32 W*H        gbuffer.bin  Uint32 W*H*4 (bit patterns of the RGBA32F texels)
//   velocity.bin Uint32 W*H*4       direct.bin   Float32 W*H*4
// or, instead of gbuffer.bin / velocity.bin, UNPACKED attribute planes the device packs (rfx_pack_gbuffer / rfx_pack_velocity):
//   aov_diffuse.bin F32 W*H*4  aov_normal.bin F32 W*H*3 (world)  aov_roughness.bin / aov_metalness.bin F32 W*H  aov_emissive.bin F32 W*H*3
//   aov_velocity.bin F32 W*H*2 (uv units)
const fs = require("fs")
const path = require("path")

function plane(file, Ctor) {
	const b = fs.readFileSync(file)
	const ab = b.buffer.slice(b.byteOffset, b.byteOffset + b.length) // own, aligned ArrayBuffer
	return new Ctor(ab)
}

function readDump(dir) {
	const meta = JSON.parse(fs.readFileSync(path.join(dir, "frame.json"), "utf8"))
	const n = meta.width * meta.height
	const frame = {
		width: meta.width,
		height: meta.height,
		camera: meta.camera,
		prevCamera: meta.prevCamera,
		depth: plane(path.join(dir, "depth.bin"), Float32Array),
		direct: plane(path.join(dir, "direct.bin"), Float32Array),
		gbuffer: null,
		velocity: null,
		aov: null
	}
	if (fs.existsSync(path.join(dir, "gbuffer.bin"))) {
		frame.gbuffer = plane(path.join(dir, "gbuffer.bin"), Uint32Array)
		frame.velocity = plane(path.join(dir, "velocity.bin"), Uint32Array)
		if (frame.gbuffer.length !== 4 * n || frame.velocity.length !== 4 * n) throw new Error("dump " + dir + ": plane sizes do not match frame.json")
	} else {
		frame.aov = {}
		for (const kc of [["diffuse", 4], ["normal", 3], ["roughness", 1], ["metalness", 1], ["emissive", 3], ["velocity", 2]]) {
			frame.aov[kc[0]] = plane(path.join(dir, "aov_" + kc[0] + ".bin"), Float32Array)
			if (frame.aov[kc[0]].length !== kc[1] * n) throw new Error("dump " + dir + ": aov_" + kc[0] + ".bin does not match frame.json")
		}
	}
	if (frame.depth.length !== n || frame.direct.length !== 4 * n) throw new Error("dump " + dir + ": plane sizes do not match frame.json")
	return frame
}

module.exports = { readDump }
