// Runs the dashboard's inline script under Node against a small DOM / canvas / fetch stub and prints what it did.
//   node run_page.js <monitor.html> <scenario.json>
// scenario: {"routes": {"/api/...": <json or {"status": 500}>}, "actions": [["click", "critical"], ["eval", "js"], ...],
//            "dump": {"text": [ids], "rows": [tbody ids], "eval": {"name": "js expression"}}}
'use strict';
const fs = require('fs');
const vm = require('vm');
const html = fs.readFileSync(process.argv[2], 'utf8');
const scenario = JSON.parse(fs.readFileSync(process.argv[3], 'utf8'));

const htmlWrites = [], errors = [], intervals = [], fetched = [];
class ClassList {
  constructor(el) { this.el = el; }
  _set() { return new Set(this.el.className.split(/\s+/).filter(Boolean)); }
  _put(s) { this.el.className = Array.from(s).join(' '); }
  add(n) { const s = this._set(); s.add(n); this._put(s); }
  remove(n) { const s = this._set(); s.delete(n); this._put(s); }
  contains(n) { return this._set().has(n); }
  toggle(n, force) { const s = this._set(); const on = force === undefined ? !s.has(n) : !!force; if (on) s.add(n); else s.delete(n); this._put(s); return on; }
}
const ctxCalls = {};
function ctx2d() {
  const c = {font: '', fillStyle: '', strokeStyle: '', lineWidth: 1, textAlign: ''};
  for (const m of ['setTransform', 'clearRect', 'beginPath', 'moveTo', 'lineTo', 'bezierCurveTo', 'stroke', 'fill', 'fillText', 'arc', 'closePath', 'setLineDash'])
    c[m] = (...a) => { ctxCalls[m] = (ctxCalls[m] || 0) + 1; for (const v of a) if (typeof v === 'number' && !isFinite(v)) errors.push('canvas ' + m + ' got ' + v); };
  return c;
}
class El {
  constructor(tag) { this.tagName = String(tag).toUpperCase(); this.children = []; this._text = ''; this.style = {}; this.className = ''; this.dataset = {};
    this.listeners = {}; this._html = null; this.parentNode = null; this.id = ''; this.title = ''; this.colSpan = 1; this.classList = new ClassList(this);
    this.width = 0; this.height = 0; this.offsetWidth = 60; this._ctx = null; }
  get textContent() { return this._text + this.children.map(c => c.textContent).join(''); }
  set textContent(v) { this.children = []; this._text = String(v); this._html = null; }
  get innerHTML() { return this._html === null ? '' : this._html; }
  set innerHTML(v) { this._html = String(v); this.children = []; this._text = ''; htmlWrites.push({id: this.id || this.className, html: String(v)}); }
  appendChild(c) { c.parentNode = this; this.children.push(c); return c; }
  addEventListener(t, fn) { (this.listeners[t] = this.listeners[t] || []).push(fn); }
  dispatch(t, ev) { (this.listeners[t] || []).forEach(fn => fn(Object.assign({target: this}, ev || {}))); }
  getBoundingClientRect() { return {width: 420, height: 180, left: 0, top: 0}; }
  getContext() { return this._ctx || (this._ctx = ctx2d()); }
}
const byId = {}, badges = [];
for (const m of html.matchAll(/<([a-zA-Z0-9]+)\b([^>]*)\bid="([^"]+)"/g)) { const e = new El(m[1]); e.id = m[3]; byId[m[3]] = e; const host = new El('div'); host.appendChild(e); }
for (const m of html.matchAll(/<button class="badge (\w+)" data-tier="(\w+)">/g)) { const e = new El('button'); e.className = 'badge ' + m[1]; e.dataset.tier = m[2]; badges.push(e); }
const document = {
  getElementById: id => { if (!byId[id]) { errors.push('no element #' + id); byId[id] = new El('div'); } return byId[id]; },
  createElement: t => new El(t),
  createTextNode: t => ({textContent: String(t), parentNode: null}),
  querySelector: sel => sel === '.badge.critical' ? badges.find(b => b.classList.contains('critical')) : null,
  querySelectorAll: sel => sel === '.badge' ? badges : [],
};
let pending = 0;
function fetchStub(url) {
  const path = String(url).split('?')[0]; fetched.push(path); pending++;
  const r = scenario.routes[path];
  return new Promise(res => setImmediate(() => { pending--;
    if (r === undefined) return res({ok: false, status: 404, json: async () => { throw new Error('404'); }});
    if (r && r.__status) return res({ok: false, status: r.__status, json: async () => ({})});
    res({ok: true, status: 200, json: async () => JSON.parse(JSON.stringify(r))}); }));
}
const sandbox = {document, console: {log() {}, error: (...a) => errors.push(a.map(x => (x && x.stack) || String(x)).join(' ')), warn() {}},
  fetch: fetchStub, setInterval: (fn, ms) => { intervals.push(ms); return intervals.length; }, setTimeout: (fn, ms) => 0, clearInterval() {},
  ResizeObserver: class { constructor(cb) { this.cb = cb; } observe() {} }, devicePixelRatio: 1};
// scenario.raf: a frame clock the test drives. requestAnimationFrame queues its callback; the action ["frames", n, stepMs] runs n
// frames, stepMs apart on performance.now() (the page animates a chart's first data over 500 ms, like the reference's Chart.js).
const rafQueue = [];
let clockMs = 1000;
if (scenario.raf) {
  sandbox.requestAnimationFrame = fn => { rafQueue.push(fn); return rafQueue.length; };
  sandbox.performance = {now: () => clockMs};
}
sandbox.window = sandbox;
vm.createContext(sandbox);
const script = html.slice(html.indexOf('<script>') + 8, html.lastIndexOf('</script>'));

function rows(id) { return (byId[id] ? byId[id].children : []).map(tr => tr.children.map(td => td.textContent)); }
async function settle() { for (let i = 0; i < 50; i++) { await new Promise(r => setImmediate(r)); if (pending === 0 && i > 5) break; } }

(async () => {
  try { vm.runInContext(script, sandbox, {filename: 'monitor.html.js'}); } catch (e) { errors.push('script threw: ' + (e.stack || e)); }
  await settle();
  for (const a of scenario.actions || []) {
    try {
      if (a[0] === 'click') badges.find(b => b.dataset.tier === a[1]).dispatch('click');
      else if (a[0] === 'eval') vm.runInContext(a[1], sandbox);
      else if (a[0] === 'mousemove') byId[a[1]].dispatch('mousemove', {clientX: a[2], clientY: 10});
      else if (a[0] === 'frames') for (let i = 0; i < a[1]; i++) { clockMs += a[2]; const q = rafQueue.splice(0); for (const fn of q) fn(clockMs); }
    } catch (e) { errors.push('action ' + JSON.stringify(a) + ' threw: ' + (e.stack || e)); }
    await settle();
  }
  const out = {errors, htmlWrites, intervals, fetched, ctxCalls, text: {}, rows: {}, evals: {}, style: {}, cls: {}};
  for (const id of (scenario.dump || {}).text || []) out.text[id] = byId[id] ? byId[id].textContent : null;
  for (const id of (scenario.dump || {}).rows || []) out.rows[id] = rows(id);
  for (const id of (scenario.dump || {}).style || []) out.style[id] = byId[id] ? byId[id].style : null;
  for (const id of (scenario.dump || {}).cls || []) out.cls[id] = byId[id] ? byId[id].className : null;
  out.badgeClasses = badges.map(b => b.className);
  for (const [k, js] of Object.entries((scenario.dump || {}).eval || {})) { try { out.evals[k] = vm.runInContext(js, sandbox); } catch (e) { out.evals[k] = 'ERR ' + e; } }
  process.stdout.write(JSON.stringify(out));
})();
