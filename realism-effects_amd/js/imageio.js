"use strict"
// On-disk outputs of an offline run (SURVEY.md §8 f4; Python twin: rfx_amd/imageio.py): the effect's final image (RFX_TEX_FINAL,
// RGBA32F, row 0 = bottom) as OpenEXR (scene-linear, float32, uncompressed scanlines), PFM, or a tone-mapped 8-bit PNG — ACES filmic,
// the reference example's renderer.toneMapping (example/main.js), then the sRGB transfer function.
const fs = require("fs")
const zlib = require("zlib")

function crc32(buf) {
	let table = crc32.table
	if (!table) {
		table = crc32.table = new Int32Array(256)
		for (let n = 0; n < 256; n++) {
			let c = n
			for (let k = 0; k < 8; k++) c = c & 1 ? 0xedb88320 ^ (c >>> 1) : c >>> 1
			table[n] = c
		}
	}
	let c = -1
	for (let i = 0; i < buf.length; i++) c = table[(c ^ buf[i]) & 255] ^ (c >>> 8)
	return (c ^ -1) >>> 0
}

// rgba8: Uint8Array(H*W*channels), row 0 = bottom
function writePNG(file, rgb8, width, height, channels) {
	const stride = width * channels
	const raw = Buffer.alloc((stride + 1) * height)
	for (let y = 0; y < height; y++) {
		raw[y * (stride + 1)] = 0 // filter type 0
		Buffer.from(rgb8.buffer, rgb8.byteOffset + (height - 1 - y) * stride, stride).copy(raw, y * (stride + 1) + 1) // PNG rows run top -> bottom
	}
	const chunk = (tag, payload) => {
		const body = Buffer.concat([Buffer.from(tag, "ascii"), payload])
		const out = Buffer.alloc(8 + payload.length + 4)
		out.writeUInt32BE(payload.length, 0)
		body.copy(out, 4)
		out.writeUInt32BE(crc32(body), 8 + payload.length)
		return out
	}
	const ihdr = Buffer.alloc(13)
	ihdr.writeUInt32BE(width, 0)
	ihdr.writeUInt32BE(height, 4)
	ihdr[8] = 8
	ihdr[9] = channels === 3 ? 2 : 6
	fs.writeFileSync(file, Buffer.concat([Buffer.from([0x89, 0x50, 0x4e, 0x47, 0x0d, 0x0a, 0x1a, 0x0a]), chunk("IHDR", ihdr), chunk("IDAT", zlib.deflateSync(raw, { level: 6 })), chunk("IEND", Buffer.alloc(0))]))
}

// ACES filmic (three.js ACESFilmicToneMapping: RRT + ODT fit, exposure / 0.6) or "linear" (clamp), then the sRGB OETF -> Uint8Array RGB
function tonemap(rgba, width, height, operator, exposure) {
	operator = operator || "aces"
	exposure = exposure === undefined ? 1 : exposure
	const out = new Uint8Array(width * height * 3)
	const san = v => (v !== v ? 0 : Math.min(Math.max(v, 0), 65504))
	const fit = v => (v * (v + 0.0245786) - 0.000090537) / (v * (0.983729 * v + 0.432951) + 0.238081)
	const oetf = v => {
		v = Math.min(Math.max(v, 0), 1)
		return Math.floor((v <= 0.0031308 ? v * 12.92 : 1.055 * Math.pow(v, 1 / 2.4) - 0.055) * 255 + 0.5)
	}
	for (let i = 0; i < width * height; i++) {
		let r = san(rgba[4 * i]) * exposure, g = san(rgba[4 * i + 1]) * exposure, b = san(rgba[4 * i + 2]) * exposure
		if (operator === "aces") {
			r /= 0.6
			g /= 0.6
			b /= 0.6
			const r1 = fit(0.59719 * r + 0.35458 * g + 0.04823 * b), g1 = fit(0.076 * r + 0.90834 * g + 0.01566 * b), b1 = fit(0.0284 * r + 0.13383 * g + 0.83777 * b)
			r = 1.60475 * r1 - 0.53108 * g1 - 0.07367 * b1
			g = -0.10208 * r1 + 1.10813 * g1 - 0.00605 * b1
			b = -0.00327 * r1 - 0.07276 * g1 + 1.07602 * b1
		}
		out[3 * i] = oetf(r)
		out[3 * i + 1] = oetf(g)
		out[3 * i + 2] = oetf(b)
	}
	return out
}

// scene-linear RGBA32F -> single-part scanline OpenEXR, FLOAT channels A B G R, no compression
function writeEXR(file, rgba, width, height) {
	const names = ["A", "B", "G", "R"], src = [3, 2, 1, 0]
	const attr = (name, type, payload) => Buffer.concat([Buffer.from(name + "\0" + type + "\0", "ascii"), (() => { const b = Buffer.alloc(4); b.writeInt32LE(payload.length, 0); return b })(), payload])
	const chlist = Buffer.concat(names.map(n => { const b = Buffer.alloc(n.length + 1 + 16); b.write(n, 0, "ascii"); b.writeInt32LE(2, n.length + 1); b.writeInt32LE(1, n.length + 9); b.writeInt32LE(1, n.length + 13); return b }).concat([Buffer.from([0])]))
	const box = Buffer.alloc(16)
	box.writeInt32LE(width - 1, 8)
	box.writeInt32LE(height - 1, 12)
	const f32 = v => { const b = Buffer.alloc(4); b.writeFloatLE(v, 0); return b }
	const header = Buffer.concat([Buffer.from([0x76, 0x2f, 0x31, 0x01, 2, 0, 0, 0]), attr("channels", "chlist", chlist), attr("compression", "compression", Buffer.from([0])),
		attr("dataWindow", "box2i", box), attr("displayWindow", "box2i", box), attr("lineOrder", "lineOrder", Buffer.from([0])), attr("pixelAspectRatio", "float", f32(1)),
		attr("screenWindowCenter", "v2f", Buffer.concat([f32(0), f32(0)])), attr("screenWindowWidth", "float", f32(1)), Buffer.from([0])])
	const lineBytes = 8 + 16 * width
	const table = Buffer.alloc(8 * height), body = Buffer.alloc(lineBytes * height)
	for (let y = 0; y < height; y++) {
		const off = header.length + table.length + y * lineBytes
		table.writeUInt32LE(off >>> 0, 8 * y)
		table.writeUInt32LE(Math.floor(off / 4294967296), 8 * y + 4)
		body.writeInt32LE(y, y * lineBytes)
		body.writeInt32LE(16 * width, y * lineBytes + 4)
		const row = height - 1 - y // EXR y = 0 is the top row
		for (let c = 0; c < 4; c++) for (let x = 0; x < width; x++) body.writeFloatLE(rgba[4 * (row * width + x) + src[c]], y * lineBytes + 8 + 4 * (c * width + x))
	}
	fs.writeFileSync(file, Buffer.concat([header, table, body]))
}

function writePFM(file, rgba, width, height) {
	const body = Buffer.alloc(12 * width * height)
	for (let i = 0; i < width * height; i++) for (let c = 0; c < 3; c++) body.writeFloatLE(rgba[4 * i + c], 4 * (3 * i + c))
	fs.writeFileSync(file, Buffer.concat([Buffer.from("PF\n" + width + " " + height + "\n-1.0\n", "ascii"), body]))
}

module.exports = { writePNG, writeEXR, writePFM, tonemap }
